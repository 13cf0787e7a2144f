"""C-ABI surface: the library loads, exports every symbol include/malio.h declares, struct layouts
match, and creating a handle without a GPU fails loudly (no CPU fallback). No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "malio.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(malio_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(capi):
    lib = capi.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/malio.h but not exported"
    assert set(syms) == set(capi.EXPORTS)


def test_option_numbers_match_the_header(capi):
    """capi.OPT (what tests, bench.py and the tools pass to malio_set_option) against include/malio.h's enum, both ways; the
    environment variables malio_create reads as initial values (csrc/capi.hip) name options that exist."""
    txt = open(os.path.join(ROOT, "include", "malio.h")).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    enum = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"\bMALIO_OPT_([A-Z0-9_]+)\s*=\s*(\d+)", code)}
    assert enum == capi.OPT and len(enum) >= 12
    src = open(os.path.join(ROOT, "ma-lio_amd", "csrc", "capi.hip")).read()
    envs = re.findall(r'\{"MALIO_([A-Z0-9_]+)",\s*MALIO_OPT_([A-Z0-9_]+)\}', src)
    assert len(envs) >= 9 and all(a == b and b.lower() in enum for a, b in envs), envs
    for name in enum:  # every option is documented where it is declared
        assert txt.count("MALIO_OPT_" + name.upper()) >= 2 or name.startswith("debug_"), name


def test_struct_layouts(capi):
    assert C.sizeof(capi.Point) == 48          # pcl::PointXYZINormal
    assert C.sizeof(capi.Pose) == 59 * 8        # common_lib.h:57-63
    assert C.sizeof(capi.State) == (3 + 4 + 4 * 4 + 4 * 3 + 12) * 8
    assert capi.Point.intensity.offset == 32 and capi.Point.normal_x.offset == 16 and capi.Point.curvature.offset == 36
    assert b"gfx950" in capi.lib().malio_version()


def test_create_without_gpu_fails_loudly(capi, scenes):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.MalioError):
        capi.Engine(scenes.DEFAULT_PARAMS, device=0)


def test_bad_args(capi):
    lib = capi.lib()
    h = C.c_void_p()
    assert lib.malio_create(None, 0, C.byref(h)) == -3
    prm = capi.make_params(dict(lid_num=9))
    assert lib.malio_create(C.byref(prm), 0, C.byref(h)) == -3
    assert lib.malio_destroy(None) == -3


def test_header_is_plain_c(tmp_path):
    """include/malio.h is the boundary a C or cgo/JNI/ctypes binding compiles against: C99, no C++ anywhere."""
    import subprocess
    src = tmp_path / "abi_c.c"
    src.write_text('#include "malio.h"\n'
                   'int main(void) { malio_xchg_t x = 0; malio_handle_t h = 0; (void)x; (void)h;\n'
                   '  return (MALIO_ERR_TIMEOUT == -7 && MALIO_SCAN_ORDER_KEEP == 2 && sizeof(malio_point_t) == 48) ? 0 : 1; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           str(src), "-o", exe])
    assert subprocess.call([exe]) == 0
