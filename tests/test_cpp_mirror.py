"""The C++ mirror of the reference interface (ma-lio_amd/host/malio_mapping.hpp) - compiled with plain g++ against
the C-ABI library, and (GPU) driven through one turn of the mapping loop, compared with the ctypes path."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "mapping_loop.cpp")


def _compile(tmp_path):
    exe = str(tmp_path / "mapping_loop")
    libdir = os.path.join(ROOT, "ma-lio_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(libdir, "host"),
           "-I" + os.path.join(ROOT, "include"), SRC, "-L" + libdir, "-lmalio_hip", "-Wl,-rpath," + libdir,
           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path, capi):
    capi.lib()  # the library must exist (built by __graft_entry__.build())
    exe = _compile(tmp_path)
    assert os.path.exists(exe)


def _dump(path, sc, orc_mod, wny):
    from oracle.orc import PARAM_ORDER
    L = sc["L"]
    with open(path, "wb") as f:
        np.array([L, sc["N"], sc["Nmap"]], np.int32).tofile(f)
        np.array([float(sc["params"][k]) for k in PARAM_ORDER[:16]], np.float64).tofile(f)
        np.asarray(sc["state0"], np.float64).tofile(f)
        np.asarray(sc["P0"], np.float64).tofile(f)
        np.ascontiguousarray(sc["map"], np.float32).tofile(f)
        np.ascontiguousarray(sc["scan"], np.float32).tofile(f)
        np.array([t.shape[0] for t in sc["tables"]], np.int32).tofile(f)
        for t in sc["tables"]:
            np.ascontiguousarray(t, np.float64).tofile(f)
        if L > 1:
            np.ascontiguousarray(sc["temporal_comp"], np.float64).tofile(f)
        wny.astype(np.float32).tofile(f)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [1, 3])
def test_cpp_mirror_mapping_loop_equals_ctypes_path(tmp_path, orc, capi, scenes, cfg):
    sc = scenes.make_scene(cfg=cfg)
    wny = np.where(np.arange(sc["N"]) < sc["N"] // 2, 0.001, 0.0).astype(np.float32)
    scene = str(tmp_path / "scene.bin")
    _dump(scene, sc, orc, wny)
    exe = _compile(tmp_path)
    out = subprocess.run([exe, scene], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = {ln.split()[0]: ln.split()[1:] for ln in out.stdout.strip().splitlines()}
    hexf = float.fromhex

    eng = capi.Engine(sc["params"])
    eng.map_build(sc["map"])
    assert int(got["size0"][0]) == eng.map_size()
    eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    r = eng.measure(sc["state0"], True, want_rows=True)
    kv = dict(zip(got["pass"][0::2], got["pass"][1::2]))
    assert int(kv["valid"]) == int(r["valid"]) and int(kv["rows"]) == r["M"] and int(kv["cols"]) == 6 * (1 + sc["L"])
    assert hexf(kv["h0"]) == r["h"][0] and hexf(kv["R0"]) == r["R"][0] and hexf(kv["HtH00"]) == r["HtRinvH"][0, 0]
    u = eng.update_iterated(sc["state0"], sc["P0"])
    vals = [hexf(v) for v in re.findall(r"-?0x[0-9a-fp.+-]+", " ".join(got["pos"]))]
    np.testing.assert_array_equal(vals[:3], u["state"][:3])
    np.testing.assert_array_equal(vals[3:7], u["state"][3:7])
    assert vals[7] == u["P"][0, 0]
    na, nn, _ = eng.map_incremental(u["state"], True, wny)
    assert int(got["add_point_size"][0]) == na + nn and int(got["add_point_size"][2]) == eng.map_size()
    pos = np.asarray(u["state"][:3], np.float32)
    deleted = eng.map_delete_boxes(np.concatenate([pos - np.float32(4), pos + np.float32(4)])[None])
    assert int(got["deleted"][0]) == deleted and int(got["deleted"][2]) == eng.map_size()
    m = eng.map_get().astype(np.float64)
    assert int(got["flatten"][0]) == m.shape[0]
    cs = 0.0
    for p in m:
        cs += p[0] + 2.0 * p[1] + 3.0 * p[2] + 1000.0 * p[5]
    assert hexf(got["flatten"][2]) == cs
    vg = eng.voxel_downsample(sc["scan"], 1.0).astype(np.float64)
    assert int(got["voxel"][0]) == vg.shape[0]
    vs = 0.0
    for p in vg:
        vs += p[0] + 2.0 * p[1] + 3.0 * p[2] + p[8]
    assert hexf(got["voxel"][2]) == vs
    Q = np.diag([0.1] * 6 + [1e-4] * 6)
    xp, Pp = capi.predict(sc["L"], u["state"], u["P"], 0.005, Q, [0.1, -0.2, 9.7], [0.01, 0.02, -0.03])
    iv = 6 * (sc["L"] + 1) + sc["L"] + 1          # flat index of vel (quaternions take 4)
    if sc["L"] > 1:
        tabs, ts = sc["tables"], 0.0
        for l in range(sc["L"]):
            for i in range(tabs[l].shape[0] - 1):
                p = tabs[l][i]
                if l > 0:
                    p = capi.compound(tabs[l][0], p)
                    p = capi.compound(sc["temporal_comp"][l - 1], p, alias=True)
                    p = capi.compound(tabs[0][0], p, inverse=True, alias=True)
                ts += p[4] + 2 * p[1] + 1e6 * p[23 + 7] + 1e6 * p[23 + 35]
        assert [int(got["tables"][0]), int(got["tables"][1])] == [tabs[0].shape[0] - 1, tabs[-1].shape[0] - 1]
        assert hexf(got["tables"][3]) == ts
    assert [hexf(v) for v in got["predict"]] == [xp[0], xp[iv + 2], Pp[0, 0], Pp[4, 5]]


@pytest.mark.gpu
@pytest.mark.parametrize("part", ["scan", "tiles"])
def test_cpp_mirror_mapping_loop_unchanged_on_a_three_shard_node(tmp_path, orc, capi, scenes, part):
    """The SAME program (tests/cpp/mapping_loop.cpp, not a line changed) with MALIO_NODE=3,<partition> in its environment:
    malio::Handle then puts three shards behind the one handle (on this box all on GPU 0) and every class of the mirror
    calls malio_node_*. Compared with ONE engine driven through ctypes: discrete results exactly (sizes, rows, counts,
    deletions, k-NN), the posterior to the summation order of the shards, the flattened map as a set."""
    sc = scenes.make_scene(cfg=1)
    wny = np.where(np.arange(sc["N"]) < sc["N"] // 2, 0.001, 0.0).astype(np.float32)
    scene = str(tmp_path / "scene.bin")
    _dump(scene, sc, orc, wny)
    exe = _compile(tmp_path)
    env = dict(os.environ, MALIO_NODE="3," + part, MALIO_NODE_SAME_DEVICE="1")
    out = subprocess.run([exe, scene], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    got = {ln.split()[0]: ln.split()[1:] for ln in out.stdout.strip().splitlines()}
    hexf = float.fromhex

    eng = capi.Engine(sc["params"])
    eng.map_build(sc["map"])
    assert int(got["size0"][0]) == eng.map_size()
    eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    r = eng.measure(sc["state0"], True)
    kv = dict(zip(got["pass"][0::2], got["pass"][1::2]))
    assert int(kv["valid"]) == int(r["valid"]) and int(kv["rows"]) == r["M"] and int(kv["cols"]) == 6 * (1 + sc["L"])
    assert abs(hexf(kv["HtH00"]) - r["HtRinvH"][0, 0]) <= 1e-12 * abs(r["HtRinvH"][0, 0])
    u = eng.update_iterated(sc["state0"], sc["P0"])
    vals = np.array([hexf(v) for v in re.findall(r"-?0x[0-9a-fp.+-]+", " ".join(got["pos"]))])
    # (the posterior covariance is the ill-conditioned end of the update: per-shard partial sums move it in the 6th digit)
    assert np.abs(vals[:7] - u["state"][:7]).max() < 1e-10 and abs(vals[7] - u["P"][0, 0]) <= 1e-4 * abs(u["P"][0, 0])
    na, nn, _ = eng.map_incremental(u["state"], True, wny)
    assert int(got["add_point_size"][0]) == na + nn and int(got["add_point_size"][2]) == eng.map_size()
    pos = np.asarray(vals[:3], np.float32)
    deleted = eng.map_delete_boxes(np.concatenate([pos - np.float32(4), pos + np.float32(4)])[None])
    if part == "scan":
        assert int(got["deleted"][0]) == deleted
    assert int(got["deleted"][2]) == eng.map_size()
    m = eng.map_get().astype(np.float64)
    assert int(got["flatten"][0]) == m.shape[0]
    cs = (m[:, 0] + 2.0 * m[:, 1] + 3.0 * m[:, 2] + 1000.0 * m[:, 5]).sum()
    assert abs(hexf(got["flatten"][2]) - cs) <= 1e-9 * abs(cs)       # (the shards hand their points over in another order)
    q = np.zeros((4, 12), np.float32)
    q[:, 0], q[:, 1], q[:, 2] = vals[0] + 6.0, vals[1], vals[2]
    _, d2, cnt = eng.nearest_search(q, 5)
    assert int(got["knn"][0]) == cnt[0] and (cnt[0] == 0 or hexf(got["knn"][2]) == float(d2[0, 0]))
