"""Named-array container shared with oracle/ref_eigen/ref_eigen_main.cpp:
record = [u32 name_len][name][u32 dtype: 0 f32, 1 f64, 2 i32][u32 ndim][u32 dims...][raw little-endian data]."""
import struct

import numpy as np

_DT = {0: np.float32, 1: np.float64, 2: np.int32}
_CODE = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int32): 2}


def write(path, arrays: dict):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<II", _CODE[a.dtype], a.ndim))
            f.write(struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(a.tobytes())


def read(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            h = f.read(4)
            if len(h) < 4:
                break
            name = f.read(struct.unpack("<I", h)[0]).decode()
            code, nd = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<%dI" % nd, f.read(4 * nd)) if nd else ()
            dt = np.dtype(_DT[code])
            n = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(n * dt.itemsize), dt).reshape(dims).copy()
    return out
