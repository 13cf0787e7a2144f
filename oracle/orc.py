"""TEST INFRASTRUCTURE - ctypes binding of the CPU oracle (oracle/liborc.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (ma-lio_amd/) never does.  See oracle/orc_capi.cpp for the flat layouts.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborc.so")
REF_SO = os.path.join(_HERE, "_ref", "libikd_ref.so")

PARAM_ORDER = [
    "lid_num", "max_iteration", "extrinsic_est_en", "plane_th", "cov_threshold", "range_min", "range_max",
    "point_cov_max", "point_cov_min", "plane_cov_max", "plane_cov_min", "localize_cov_max", "localize_cov_min",
    "localize_thresh_max", "localize_thresh_min", "filter_size_map", "limit",
]


def build(force=False):
    """Compile the restatement (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or os.path.exists("/root/reference/MA_LIO"):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_spline_create.restype = C.c_void_p
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def state_len(L):
    return 19 + 7 * L


class Oracle:
    """One Scene (the globals h_share_model touches) plus its k-NN provider."""

    def __init__(self, params: dict, threads=1, use_ref=False):
        prm = np.array([float(params.get(k, 0.0)) for k in PARAM_ORDER], dtype=np.float64)  # limit: 0 = 0.001
        self.params = dict(params)
        self.L = int(params["lid_num"])
        self.C = 6 * (1 + self.L)
        self.n = 17 + 6 * self.L
        ref = REF_SO.encode() if (use_ref and have_ref()) else b""
        self.h = C.c_void_p(lib().orc_create(_p(prm, C.c_double), int(threads), ref))
        self.is_ref = bool(lib().orc_is_ref(self.h))
        self.N = 0

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_replay(self, passes):
        """passes: list of dict(valid, h_x [M,C], h [M], R [M]) handed back by successive h_share_model calls ([] = off)."""
        n = len(passes)
        valid = np.array([int(p["valid"]) for p in passes] or [0], np.int32)
        M = np.array([p["h_x"].shape[0] for p in passes] or [0], np.int32)
        hx = _f64(np.concatenate([p["h_x"] for p in passes])) if n else np.zeros((1, self.C))
        hv = _f64(np.concatenate([p["h"] for p in passes])) if n else np.zeros(1)
        Rv = _f64(np.concatenate([p["R"] for p in passes])) if n else np.zeros(1)
        lib().orc_set_replay(self.h, n, _p(valid, C.c_int), _p(M, C.c_int), _p(hx, C.c_double), _p(hv, C.c_double),
                             _p(Rv, C.c_double))

    def set_pass_hook(self, fn):
        """fn(pass_number) is called before every measurement pass of update_iterated (None removes it)."""
        self._hook = C.CFUNCTYPE(None, C.c_int, C.c_void_p)(lambda k, _u: fn(k)) if fn else None
        lib().orc_set_pass_hook(self.h, self._hook if fn else C.cast(None, C.CFUNCTYPE(None, C.c_int, C.c_void_p)), None)

    def set_threads(self, t):
        lib().orc_set_threads(self.h, int(t))

    def map_build(self, pts12):
        pts12 = _f32(pts12)
        return lib().orc_map_build(self.h, _p(pts12, C.c_float), pts12.shape[0])

    def knn(self, q12, k=5):
        q12 = _f32(q12)
        n = q12.shape[0]
        out = np.zeros((n, k, 12), np.float32)
        d2 = np.zeros((n, k), np.float32)
        cnt = np.zeros(n, np.int32)
        lib().orc_knn(self.h, _p(q12, C.c_float), n, k, _p(out, C.c_float), _p(d2, C.c_float), _p(cnt, C.c_int))
        return out, d2, cnt

    def scan_set(self, pts12, pose_tables, temporal_comp):
        """pose_tables: list (per lidar) of [k,59] arrays; temporal_comp: [L-1,59]."""
        pts12 = _f32(pts12)
        self.N = pts12.shape[0]
        lens = np.array([t.shape[0] for t in pose_tables], np.int32)
        tables = _f64(np.concatenate([np.asarray(t, np.float64).reshape(-1, 59) for t in pose_tables], 0))
        tc = _f64(np.asarray(temporal_comp, np.float64).reshape(-1, 59)) if self.L > 1 else np.zeros((1, 59))
        lib().orc_scan_set(self.h, _p(pts12, C.c_float), self.N, _p(lens, C.c_int), _p(tables, C.c_double),
                           _p(tc, C.c_double))

    def h_share_model(self, state, converge=True):
        state = _f64(state)
        hx = np.zeros((self.N, self.C), np.float64)
        hv = np.zeros(self.N, np.float64)
        Rv = np.zeros(self.N, np.float64)
        valid = C.c_int(0)
        w = C.c_double(0)
        M = lib().orc_h_share_model(self.h, _p(state, C.c_double), int(bool(converge)), C.byref(valid),
                                    _p(hx, C.c_double), _p(hv, C.c_double), _p(Rv, C.c_double), C.byref(w))
        return dict(valid=bool(valid.value), M=M, h_x=hx[:M].copy(), h=hv[:M].copy(), R=Rv[:M].copy(),
                    weight=w.value)

    def last_minmax(self):
        out = np.zeros(4, np.float64)
        lib().orc_last_minmax(self.h, _p(out, C.c_double))
        return out

    def set_override(self, mm4=None, skip_loc_weight=False):
        mm = _f64(mm4) if mm4 is not None else None
        lib().orc_set_override(self.h, _p(mm, C.c_double) if mm is not None else None, int(bool(skip_loc_weight)))

    def scan_get(self):
        n = self.N
        out = dict(normal_y=np.zeros(n, np.float32), nearest=np.zeros((n, 5, 12), np.float32),
                   nearest_cnt=np.zeros(n, np.int32), selected=np.zeros(n, np.uint8),
                   res_last=np.zeros(n, np.float32), world=np.zeros((n, 3), np.float32),
                   normvec=np.zeros((n, 4), np.float32))
        lib().orc_scan_get(self.h, _p(out["normal_y"], C.c_float), _p(out["nearest"], C.c_float),
                           _p(out["nearest_cnt"], C.c_int), _p(out["selected"], C.c_ubyte),
                           _p(out["res_last"], C.c_float), _p(out["world"], C.c_float),
                           _p(out["normvec"], C.c_float))
        return out

    def map_incremental(self, state, flg_EKF_inited=True, world_normal_y=None):
        """laserMapping.cpp:398-442: returns (PointToAdd [na,12], PointNoNeedDownsample [nn,12])."""
        state = _f64(state)
        a = np.zeros((max(self.N, 1), 12), np.float32)
        b = np.zeros((max(self.N, 1), 12), np.float32)
        cnt = np.zeros(2, np.int32)
        wny = _f32(world_normal_y) if world_normal_y is not None else None
        lib().orc_map_incremental(self.h, _p(state, C.c_double), int(bool(flg_EKF_inited)),
                                  _p(wny, C.c_float) if wny is not None else None, _p(a, C.c_float), _p(b, C.c_float),
                                  _p(cnt, C.c_int))
        return a[:cnt[0]].copy(), b[:cnt[1]].copy()

    def update_iterated(self, state, P, R=0.001):
        state = _f64(state).copy()
        P = _f64(P).copy()
        stats = np.zeros(3, np.int32)
        K = int(self.params["max_iteration"]) + 1
        trace = np.zeros((K, state_len(self.L)), np.float64)
        st = C.c_double(0)
        nt = lib().orc_update_iterated(self.h, _p(state, C.c_double), _p(P, C.c_double), C.c_double(R),
                                       _p(stats, C.c_int), _p(trace, C.c_double), C.byref(st))
        return dict(state=state, P=P, passes=int(stats[0]), searches=int(stats[1]), M=int(stats[2]),
                    trace=trace[:nt].copy(), solve_time=st.value)


def predict(L, state, P, dt, Q, acc, gyro):
    """esekfom.hpp:388-492 + use-ikfom.hpp:67-112 (dense, as the reference forms it). Returns (state, P)."""
    state = np.array(state, np.float64)
    P = np.array(P, np.float64, order="C")
    lib().orc_predict(int(L), _p(state, C.c_double), _p(P, C.c_double), C.c_double(dt), _p(_f64(Q), C.c_double),
                      _p(_f64(acc), C.c_double), _p(_f64(gyro), C.c_double))
    return state, P


def esti_plane(near12, threshold, cov_threshold):
    near12 = _f32(near12)
    pabcd = np.zeros(4, np.float32)
    pc = C.c_double(0)
    ok = lib().orc_esti_plane(_p(near12, C.c_float), C.c_float(threshold), C.c_double(cov_threshold),
                              _p(pabcd, C.c_float), C.byref(pc))
    return bool(ok), pabcd, pc.value


def eval_point_uncertainty(p12, pose59):
    p12 = _f32(p12)
    pose59 = _f64(pose59)
    cov = np.zeros((3, 3), np.float64)
    lib().orc_eval_point_uncertainty(_p(p12, C.c_float), _p(pose59, C.c_double), _p(cov, C.c_double))
    return cov


def compound(pose1, pose2, inverse=False, alias=False):
    out = np.zeros(59, np.float64)
    lib().orc_compound(_p(_f64(pose1), C.c_double), _p(_f64(pose2), C.c_double), int(inverse), int(alias),
                       _p(out, C.c_double))
    return out


def boxplus(state, L, dx):
    s = _f64(state).copy()
    lib().orc_boxplus(_p(s, C.c_double), int(L), _p(_f64(dx), C.c_double))
    return s


def boxminus(state, other, L):
    res = np.zeros(17 + 6 * L, np.float64)
    lib().orc_boxminus(_p(_f64(state), C.c_double), _p(_f64(other), C.c_double), int(L), _p(res, C.c_double))
    return res


class Spline:
    def __init__(self, traj8):
        traj8 = _f64(traj8)
        self.h = C.c_void_p(lib().orc_spline_create(_p(traj8, C.c_double), traj8.shape[0]))

    def __del__(self):
        try:
            if self.h:
                lib().orc_spline_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def control(self):
        n = lib().orc_spline_num_control(self.h)
        t = np.zeros(n, np.float64)
        T = np.zeros((n, 4, 4), np.float64)
        lib().orc_spline_control(self.h, _p(t, C.c_double), _p(T, C.c_double))
        return t, T

    def get_pose(self, t):
        q = np.zeros(4, np.float64)
        p = np.zeros(3, np.float64)
        ok = lib().orc_spline_get_pose(self.h, C.c_double(t), _p(q, C.c_double), _p(p, C.c_double))
        return bool(ok), q, p

    def undistort(self, pts12, beg, end, imu_t, imu_cov, ext59, lt59, cap=256):
        pts = _f32(pts12).copy()
        imu_t = _f64(imu_t)
        imu_cov = _f64(imu_cov)
        unc = np.zeros((cap, 59), np.float64)
        n = lib().orc_undistort(self.h, _p(pts, C.c_float), pts.shape[0], C.c_double(beg), C.c_double(end),
                                _p(imu_t, C.c_double), _p(imu_cov, C.c_double), imu_t.shape[0],
                                _p(_f64(ext59), C.c_double), _p(_f64(lt59), C.c_double), _p(unc, C.c_double), cap)
        return pts, unc[:n].copy()


class VoxMap:
    """Flat-list restatement of Add_Points / Delete_Point_Boxes (oracle/orc_map.cpp)."""

    def __init__(self, downsample):
        lib().orc_vmap_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_vmap_create(C.c_float(downsample)))
        self._pfx = "orc_vmap_"
        self._l = lib()

    def _f(self, name):
        return getattr(self._l, self._pfx + name)

    def __del__(self):
        try:
            if self.h:
                self._f("destroy")(self.h)
                self.h = None
        except Exception:
            pass

    def build(self, pts12):
        pts12 = _f32(pts12)
        self._f("build")(self.h, _p(pts12, C.c_float), pts12.shape[0])

    def size(self):
        return self._f("size")(self.h)

    def add(self, pts12, downsample_on=True):
        pts12 = _f32(pts12).reshape(-1, 12)
        return self._f("add")(self.h, _p(pts12, C.c_float), pts12.shape[0], int(bool(downsample_on)))

    def delete_boxes(self, boxes6):
        boxes6 = _f32(boxes6).reshape(-1, 6)
        return self._f("delete_boxes")(self.h, _p(boxes6, C.c_float), boxes6.shape[0])

    def flatten(self):
        n = self.size()
        out = np.zeros((max(n, 1), 12), np.float32)
        got = self._f("flatten")(self.h, _p(out, C.c_float), n)
        assert got == n
        return out[:n]


class RefTree(VoxMap):
    """The REFERENCE's own ikd-Tree (oracle/_ref/libikd_ref.so, compiled from /root/reference in place)."""

    def __init__(self, downsample):
        if not have_ref():
            raise RuntimeError("oracle/_ref/libikd_ref.so not built")
        self._l = C.CDLL(REF_SO)
        self._l.refikd_create.restype = C.c_void_p
        self._pfx = "refikd_"
        self.h = C.c_void_p(self._l.refikd_create(C.c_float(downsample)))

    def size(self):
        return self._l.refikd_validnum(self.h)


def voxel_downsample(pts12, leaf, normalize_normal=True):
    """Restated pcl::VoxelGrid (oracle/orc_voxel.cpp): [n,12] -> [n_voxels,12] in ascending voxel-index order."""
    pts12 = _f32(pts12).reshape(-1, 12)
    n = pts12.shape[0]
    out = np.zeros((max(n, 1), 12), np.float32)
    m = lib().orc_voxel_downsample(_p(pts12, C.c_float), n, C.c_float(leaf), int(bool(normalize_normal)),
                                   _p(out, C.c_float), n)
    return out[:m].copy()


def decode_livox(records, n_scans, point_filter_num, blind, eof_point=False):
    """Restated file_player reader + Preprocess::avia_handler (oracle/orc_decode.cpp)."""
    rec = np.frombuffer(bytes(records), np.uint8) if not isinstance(records, np.ndarray) else np.ascontiguousarray(records, np.uint8)
    n = rec.size // 19
    out = np.zeros((n + 2, 12), np.float32)
    mt = C.c_double(0)
    m = lib().orc_decode_livox(rec.ctypes.data_as(C.POINTER(C.c_ubyte)), n, int(n_scans), int(point_filter_num),
                               C.c_double(blind), int(bool(eof_point)), _p(out, C.c_float), n + 2, C.byref(mt))
    return out[:m].copy(), mt.value


def decode_ouster(records, point_filter_num, blind, time_unit_scale):
    """Restated file_player reader + Preprocess::oust64_handler (oracle/orc_decode.cpp)."""
    rec = np.frombuffer(bytes(records), np.uint8) if not isinstance(records, np.ndarray) else np.ascontiguousarray(records, np.uint8)
    n = rec.size // 22
    out = np.zeros((n + 1, 12), np.float32)
    mt = C.c_double(0)
    m = lib().orc_decode_ouster(rec.ctypes.data_as(C.POINTER(C.c_ubyte)), n, int(point_filter_num), C.c_double(blind),
                                C.c_float(time_unit_scale), _p(out, C.c_float), n + 1, C.byref(mt))
    return out[:m].copy(), mt.value


def decode_velodyne(data, n_points, layout, point_filter_num, blind, time_unit_scale, maximum_time_in=-1.0):
    """Restated pcl::fromROSMsg + Preprocess::velodyne_handler (oracle/orc_decode.cpp). layout = (point_step, off_x, off_y,
    off_z, off_intensity, off_time); maximum_time_in: the member's value before the call (kept when n_points == 0)."""
    rec = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    n = int(n_points)
    out = np.zeros((n + 1, 12), np.float32)
    mt = C.c_double(maximum_time_in)
    m = lib().orc_decode_velodyne(rec.ctypes.data_as(C.POINTER(C.c_ubyte)), n, *[int(v) for v in layout], int(point_filter_num),
                                  C.c_double(blind), C.c_float(time_unit_scale), _p(out, C.c_float), n + 1, C.byref(mt))
    return out[:m].copy(), mt.value
