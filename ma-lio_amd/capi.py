"""ctypes view of the C ABI in include/malio.h (libmalio_hip.so).

This is plumbing for tests and bench.py: the product is the shared library. Loading fails loudly
when the library has not been built (`__graft_entry__.build()` / `make -C ma-lio_amd`); creating a
handle fails with MALIO_ERR_NO_DEVICE when no gfx950 GPU is visible - there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmalio_hip.so")

MAX_LIDAR = 4
OK, NO_EFFECTIVE_POINTS, SMALL_M_FALLBACK = 0, 1, 2
ERR_NO_DEVICE = -1
ERR_BAD_ARG = -3
ERR_TIMEOUT = -7

EXPORTS = [
    "malio_create", "malio_destroy", "malio_version", "malio_build_id", "malio_device_count", "malio_last_error", "malio_set_stream", "malio_map_build",
    "malio_map_size", "malio_nearest_search", "malio_map_add", "malio_map_delete_boxes", "malio_map_get", "malio_map_incremental", "malio_map_incremental_select", "malio_node_map_incremental", "malio_node_undistort_resident", "malio_node_scan_set_resident", "malio_node_nearest_search", "malio_node_map_get", "malio_node_map_total", "malio_node_voxel_downsample", "malio_decode_livox", "malio_decode_ouster", "malio_decode_velodyne", "malio_voxel_downsample", "malio_undistort_resident", "malio_scan_set_resident", "malio_scan_set", "malio_scan_set_packed", "malio_scan_upload_wait", "malio_scan_stage",
    "malio_measure", "malio_scan_get", "malio_update_iterated", "malio_update_iterated_begin", "malio_update_iterated_end", "malio_undistort", "malio_sums_len",
    "malio_measure_stage1", "malio_measure_stage2", "malio_measure_finish", "malio_last_kernel_times",
    "malio_set_profiling", "malio_set_partition", "malio_set_partition_shape", "malio_part_owner_shape", "malio_part_stores_shape", "malio_scan_owned", "malio_set_pass_hook", "malio_ieskf_step", "malio_predict", "malio_host_alloc", "malio_host_free", "malio_result_buffer", "malio_scan_order", "malio_measure_stage2_emit", "malio_xchg_create", "malio_xchg_all_gather", "malio_xchg_reduce", "malio_xchg_row", "malio_measure_node", "malio_node_stats", "malio_update_iterated_node", "malio_xchg_unlink", "malio_xchg_destroy", "malio_debug_counters", "malio_debug_fuse_stats", "malio_debug_nfound_hist", "malio_spline_feed", "malio_spline_get_pose",
    "malio_compound_pose_cov", "malio_compound_inv_pose_cov", "malio_eval_point_uncertainty",
    "malio_xchg_create_local", "malio_debug_xchg_latency", "malio_rccl_unique_id", "malio_xchg_create_rccl", "malio_xchg_device_row", "malio_xchg_kind",
    "malio_xchg_reduce_stream", "malio_node_create", "malio_node_destroy", "malio_node_last_error", "malio_node_gpus",
    "malio_node_handle", "malio_node_map_build", "malio_node_map_size", "malio_node_map_add", "malio_node_map_delete_boxes",
    "malio_node_scan_set", "malio_node_measure", "malio_node_update_iterated", "malio_node_scan_get",
    "malio_node_set_pass_hook", "malio_node_exchange_stats", "malio_node_update_stats", "malio_part_owner", "malio_part_stores",
    "malio_set_update_mode", "malio_localize_weight", "malio_predict_chain",
    "malio_set_option", "malio_get_option", "malio_debug_skip_stats", "malio_debug_list_order", "malio_node_set_option",
]
# malio_set_option (include/malio.h)
OPT = dict(fuse=1, search_skip=2, maint_stream=3, mapinc_small=4, gate_pinned=5, gate_timeout_ms=6, scan_set_sync=7,
           nl_full_blocks=8, node_gated=9, nl_sorted=10, probe_cache=11, early_min_queries=12, map_cell_order=13, debug_fuse_bad_guess=100, debug_gate_stall_ms=101, debug_node_gated_runs=102,
           debug_node_gated_redone=103)
PART_SCAN, PART_TILES, PART_COLUMNS = 0, 1, 2  # (COLUMNS: tiles that are whole vertical columns, no halo above / below)
TILE_CUBES, TILE_COLUMNS = 0, 1
XCHG_HOST, XCHG_RCCL = 0, 1


class Point(C.Structure):  # malio_point_t == pcl::PointXYZINormal
    _fields_ = [(n, C.c_float) for n in ("x", "y", "z", "_pad0", "normal_x", "normal_y", "normal_z", "_pad1",
                                         "intensity", "curvature", "_pad2", "_pad3")]


class Pc2Layout(C.Structure):  # malio_pc2_layout_t
    _fields_ = [(n, C.c_int) for n in ("point_step", "off_x", "off_y", "off_z", "off_intensity", "off_time")]


class Pose(C.Structure):  # malio_pose_t
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("T", C.c_double * 16), ("cov", C.c_double * 36)]


class Params(C.Structure):  # malio_params_t
    _fields_ = [("lid_num", C.c_int32), ("max_iteration", C.c_int32), ("extrinsic_est_en", C.c_int32),
                ("plane_th", C.c_float), ("cov_threshold", C.c_double), ("range_min", C.c_double),
                ("range_max", C.c_double), ("point_cov_max", C.c_double), ("point_cov_min", C.c_double),
                ("plane_cov_max", C.c_double), ("plane_cov_min", C.c_double), ("localize_cov_max", C.c_double),
                ("localize_cov_min", C.c_double), ("localize_thresh_max", C.c_double),
                ("localize_thresh_min", C.c_double), ("filter_size_map", C.c_double), ("cell_size", C.c_float),
                ("reserved", C.c_int32), ("limit", C.c_double)]


class State(C.Structure):  # malio_state_t
    _fields_ = [("pos", C.c_double * 3), ("rot", C.c_double * 4), ("offset_R", (C.c_double * 4) * MAX_LIDAR),
                ("offset_T", (C.c_double * 3) * MAX_LIDAR), ("vel", C.c_double * 3), ("bg", C.c_double * 3),
                ("ba", C.c_double * 3), ("grav", C.c_double * 3)]


class MeasureOut(C.Structure):  # malio_measure_out_t
    _fields_ = [("valid", C.c_int32), ("M", C.c_int32), ("w_loc", C.c_double), ("unit_cov_minmax", C.c_double * 2),
                ("R_minmax", C.c_double * 2), ("HtRinvH", C.c_double * (36 * (1 + MAX_LIDAR) ** 2)),
                ("HtRinvh", C.c_double * (6 * (1 + MAX_LIDAR))), ("h_x", C.POINTER(C.c_double)),
                ("h", C.POINTER(C.c_double)), ("R", C.POINTER(C.c_double))]


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime and one RCCL per process. PyTorch-ROCm wheels bundle their own libamdhip64 (soname
    libamdhip64.so.7) and librccl (librccl.so.1), looked up by FILE name; libmalio_hip.so links both by soname. If
    /opt/rocm's copies are mapped first, torch later maps its own next to them and its device init fails ("No HIP GPUs
    are available"). Mapping torch's copies first works for HIP alone, but with RCCL mapped before the rest of torch the
    process dies in a static destructor at exit ("double free or corruption", MI355X box, r02b) - the only order that is
    clean in every combination is torch's own: import it first when it is installed, and let the loader satisfy our
    DT_NEEDED entries with what it mapped. No torch -> the system libraries (a C++ integration never meets this)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations or os.environ.get("MALIO_TORCH_FIRST", "1") == "0":
        return  # (MALIO_TORCH_FIRST=0: a process that will never import torch, e.g. the host-only exchange workers)
    try:
        import torch  # noqa: F401
    except Exception:
        return
    for name in ("libamdhip64.so", "librccl.so"):  # mapped by torch lazily in some builds: make sure before we load
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("MALIO_LIB", LIB_PATH)  # developer override: A/B runs of kernel variants
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with __graft_entry__.build() "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _share_hip_runtime_with_torch()
        _lib = C.CDLL(path)
        _lib.malio_version.restype = C.c_char_p
        _lib.malio_build_id.restype = C.c_char_p
        _lib.malio_last_error.restype = C.c_char_p
        _lib.malio_last_error.argtypes = [C.c_void_p]
        _lib.malio_localize_weight.restype = C.c_double
        _lib.malio_localize_weight.argtypes = [C.POINTER(C.c_double)] + [C.c_double] * 4
        for name in EXPORTS:
            getattr(_lib, name)  # AttributeError if an include/malio.h symbol is not exported
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def state_from_flat(flat, L):
    """[19+7L] flat state (scenes.pack_state layout) -> malio_state_t."""
    flat = np.asarray(flat, np.float64)
    s = State()
    o = 0
    s.pos[:] = flat[o:o + 3]; o += 3
    s.rot[:] = flat[o:o + 4]; o += 4
    for l in range(MAX_LIDAR):
        s.offset_R[l][:] = [0, 0, 0, 1]
    for l in range(L):
        s.offset_R[l][:] = flat[o:o + 4]; o += 4
    for l in range(L):
        s.offset_T[l][:] = flat[o:o + 3]; o += 3
    s.vel[:] = flat[o:o + 3]; o += 3
    s.bg[:] = flat[o:o + 3]; o += 3
    s.ba[:] = flat[o:o + 3]; o += 3
    s.grav[:] = flat[o:o + 3]
    return s


def state_to_flat(s, L):
    out = list(s.pos) + list(s.rot)
    for l in range(L):
        out += list(s.offset_R[l])
    for l in range(L):
        out += list(s.offset_T[l])
    out += list(s.vel) + list(s.bg) + list(s.ba) + list(s.grav)
    return np.array(out, np.float64)


def make_params(p: dict, cell_size=0.0):
    prm = Params()
    for k, _ in Params._fields_:
        if k in p:
            setattr(prm, k, p[k])
    prm.cell_size = cell_size
    return prm


class MalioError(RuntimeError):
    pass


def _scan_tables(L, pose_tables, temporal_comp):
    """ctypes views of pose_unc[lid][k] / temporal_comp (+ the arrays that must outlive the call)."""
    tabs = [np.ascontiguousarray(np.asarray(t, np.float64).reshape(-1, 59)) for t in pose_tables]
    ptrs = (C.POINTER(Pose) * L)(*[_p(t, Pose) for t in tabs])
    lens = (C.c_int * L)(*[t.shape[0] for t in tabs])
    tc = np.ascontiguousarray(np.asarray(temporal_comp, np.float64).reshape(-1, 59))
    tcp = _p(tc, Pose) if L > 1 else None
    return ptrs, lens, tcp, tabs, tc


class Engine:
    """One libmalio_hip handle (one GPU, one stream)."""

    def __init__(self, params: dict, device=0, cell_size=0.0):
        self.L = int(params["lid_num"])
        self.C = 6 * (1 + self.L)
        self.n = 17 + 6 * self.L
        self.params = dict(params)
        self._prm = make_params(params, cell_size)
        self.h = C.c_void_p()
        rc = lib().malio_create(C.byref(self._prm), int(device), C.byref(self.h))
        if rc != OK:
            raise MalioError(f"malio_create failed rc={rc} (no gfx950 device? there is no CPU fallback)")
        self.N = 0

    def _chk(self, rc, what):
        if rc < 0:
            raise MalioError(f"{what} rc={rc}: {lib().malio_last_error(self.h).decode()}")
        return rc

    def close(self):
        if getattr(self, "h", None):
            lib().malio_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr, external=True):
        """external=True: run on the caller's stream (0 = legacy default stream, PyTorch's default)."""
        self._chk(lib().malio_set_stream(self.h, C.c_void_p(stream_ptr), int(bool(external))), "malio_set_stream")

    def set_profiling(self, on=True):
        self._chk(lib().malio_set_profiling(self.h, int(on)), "malio_set_profiling")

    def set_update_mode(self, mode):
        """"device" (default): update_iterated is one enqueued chain of kernels with the filter algebra on the GPU;
        "host": one pass at a time, algebra on the calling thread."""
        self._chk(lib().malio_set_update_mode(self.h, {"device": 0, "host": 1, "gated": 2}[mode]), "malio_set_update_mode")

    def set_option(self, name, value):
        """malio_set_option: name is a key of OPT (or the numeric id)."""
        f = lib().malio_set_option
        f.argtypes = [C.c_void_p, C.c_int, C.c_double]
        self._chk(f(self.h, int(OPT.get(name, name)), float(value)), "malio_set_option(%s)" % name)
        return self

    def get_option(self, name):
        f = lib().malio_get_option
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        v = C.c_double(0)
        self._chk(f(self.h, int(OPT.get(name, name)), C.byref(v)), "malio_get_option(%s)" % name)
        return v.value

    def skip_stats(self):
        """After a search pass: points, points that kept their cached neighbours, points that walked the lists, allowed."""
        out = (C.c_int * 4)()
        self._chk(lib().malio_debug_skip_stats(self.h, out), "malio_debug_skip_stats")
        return dict(points=out[0], kept=out[1], walked=out[2], allowed=out[3])

    def list_order(self):
        """The level-1 neighbour lists as the next search finds them: lists, flagged ordered, flagged but NOT ordered, entries."""
        out = (C.c_longlong * 4)()
        self._chk(lib().malio_debug_list_order(self.h, out), "malio_debug_list_order")
        return dict(lists=out[0], ordered=out[1], broken=out[2], entries=out[3])

    def last_kernel_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = C.c_int(0)
        lib().malio_last_kernel_times(self.h, names, ms, 16, C.byref(n))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def debug_counters(self):
        out = (C.c_int * 8)()
        lib().malio_debug_counters(self.h, out)
        return dict(nl1_cells=out[0], map_points=out[1], nl2_cells=out[2], rebuilds=out[3], inplace=out[4],
                    dead_slots=out[5], tombstones=out[6], slots=out[7])

    def nfound_hist(self):
        out = (C.c_int * 8)()
        self._chk(lib().malio_debug_nfound_hist(self.h, out), "malio_debug_nfound_hist")
        return list(out)

    def fuse_stats(self):
        out = (C.c_int * 4)()
        lib().malio_debug_fuse_stats(self.h, out)
        return dict(passes=out[0], hits=out[1], misses=out[2], gate_timeouts=out[3])

    def map_build(self, pts12):
        pts12 = np.ascontiguousarray(pts12, np.float32)
        self._chk(lib().malio_map_build(self.h, _p(pts12, Point), pts12.shape[0]), "malio_map_build")

    def map_add(self, pts12, downsample_on=True):
        """ikdtree.Add_Points: returns the reference's return value (insertions performed)."""
        pts12 = np.ascontiguousarray(pts12, np.float32).reshape(-1, 12)
        added = C.c_int(0)
        self._chk(lib().malio_map_add(self.h, _p(pts12, Point), pts12.shape[0], int(bool(downsample_on)),
                                      C.byref(added)), "malio_map_add")
        return added.value

    def map_delete_boxes(self, boxes6):
        """ikdtree.Delete_Point_Boxes: boxes6 [nb, 6] = vertex_min xyz, vertex_max xyz. Returns #deleted."""
        boxes6 = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
        deleted = C.c_int(0)
        self._chk(lib().malio_map_delete_boxes(self.h, boxes6.ctypes.data_as(C.c_void_p), boxes6.shape[0],
                                               C.byref(deleted)), "malio_map_delete_boxes")
        return deleted.value

    def map_incremental(self, state_flat, flg_EKF_inited=True, world_normal_y=None):
        """laserMapping.cpp:398-446 on the resident scan. Returns (|PointToAdd|, |PointNoNeedDownsample|,
        Add_Points(PointToAdd, true) return value)."""
        s = state_from_flat(state_flat, self.L)
        cnt = (C.c_int * 3)()
        wny = None
        if world_normal_y is not None:
            wny = np.ascontiguousarray(world_normal_y, np.float32)
            assert wny.shape[0] == self.N
        self._chk(lib().malio_map_incremental(self.h, C.byref(s), int(bool(flg_EKF_inited)), _p(wny, C.c_float),
                                              cnt), "malio_map_incremental")
        return cnt[0], cnt[1], cnt[2]

    def map_incremental_fn(self, world_normal_y=None, flg_EKF_inited=True):
        """malio_map_incremental with its ctypes arguments built beforehand: returns (call, counts) where call(state_struct)
        is the bare C call (bench.py times the call, not Python's marshalling) and counts the int[3] it fills."""
        cnt = (C.c_int * 3)()
        wny = None if world_normal_y is None else np.ascontiguousarray(world_normal_y, np.float32)
        wp = _p(wny, C.c_float)
        fn = lib().malio_map_incremental
        flg = int(bool(flg_EKF_inited))

        def call(state_struct):
            self._keep_wny = wny
            return self._chk(fn(self.h, C.byref(state_struct), flg, wp, cnt), "malio_map_incremental")
        return call, cnt

    def map_get(self):
        """ikdtree.flatten: [n, 12] valid map points (x, y, z, normal_y populated)."""
        n = self.map_size()
        out = np.zeros((max(n, 1), 12), np.float32)
        got = C.c_int(0)
        self._chk(lib().malio_map_get(self.h, _p(out, Point), n, C.byref(got)), "malio_map_get")
        return out[:n]

    def decode_livox(self, records, n_scans, point_filter_num, blind, eof_point=False):
        """19-byte Livox records (bytes / uint8 array) -> (pl_surf [m,12], maximum_time), Preprocess::avia_handler."""
        rec = np.frombuffer(bytes(records), np.uint8) if not isinstance(records, np.ndarray) else np.ascontiguousarray(records, np.uint8)
        n = rec.size // 19
        out = np.zeros((n + 2, 12), np.float32)
        m, mt = C.c_int(0), C.c_double(0)
        self._chk(lib().malio_decode_livox(self.h, rec.ctypes.data_as(C.POINTER(C.c_ubyte)), n, int(n_scans),
                                           int(point_filter_num), C.c_double(blind), int(bool(eof_point)), _p(out, Point),
                                           n + 2, C.byref(m), C.byref(mt)), "malio_decode_livox")
        return out[:m.value].copy(), mt.value

    def decode_ouster(self, records, point_filter_num, blind, time_unit_scale):
        """22-byte Ouster records -> (pl_surf [m,12], maximum_time), Preprocess::oust64_handler."""
        rec = np.frombuffer(bytes(records), np.uint8) if not isinstance(records, np.ndarray) else np.ascontiguousarray(records, np.uint8)
        n = rec.size // 22
        out = np.zeros((n + 1, 12), np.float32)
        m, mt = C.c_int(0), C.c_double(0)
        self._chk(lib().malio_decode_ouster(self.h, rec.ctypes.data_as(C.POINTER(C.c_ubyte)), n, int(point_filter_num),
                                            C.c_double(blind), C.c_float(time_unit_scale), _p(out, Point), n + 1,
                                            C.byref(m), C.byref(mt)), "malio_decode_ouster")
        return out[:m.value].copy(), mt.value

    def decode_velodyne(self, data, n_points, layout, point_filter_num, blind, time_unit_scale, maximum_time_in=-1.0):
        """PointCloud2 data[] of a Velodyne message -> (pl_surf [m,12], maximum_time), Preprocess::velodyne_handler.
        layout = (point_step, off_x, off_y, off_z, off_intensity, off_time) in bytes (< 0: field absent)."""
        rec = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
        n = int(n_points)
        out = np.zeros((n + 1, 12), np.float32)
        lay = Pc2Layout(*[int(v) for v in layout])
        m, mt = C.c_int(0), C.c_double(maximum_time_in)
        self._chk(lib().malio_decode_velodyne(self.h, rec.ctypes.data_as(C.POINTER(C.c_ubyte)), n, C.byref(lay),
                                              int(point_filter_num), C.c_double(blind), C.c_float(time_unit_scale),
                                              _p(out, Point), n + 1, C.byref(m), C.byref(mt)), "malio_decode_velodyne")
        return out[:m.value].copy(), mt.value

    def voxel_downsample(self, pts12, leaf, normal_mode=1):
        """pcl::VoxelGrid (all fields) as restated in csrc/voxel.hip: [n,12] -> [n_voxels,12] in voxel-index order."""
        pts12 = np.ascontiguousarray(pts12, np.float32).reshape(-1, 12)
        n = pts12.shape[0]
        out = np.zeros((max(n, 1), 12), np.float32)
        got = C.c_int(0)
        self._chk(lib().malio_voxel_downsample(self.h, _p(pts12, Point), n, C.c_float(leaf), int(normal_mode),
                                               _p(out, Point), n, C.byref(got)), "malio_voxel_downsample")
        return out[:got.value].copy()

    def map_size(self):
        n = C.c_int(0)
        self._chk(lib().malio_map_size(self.h, C.byref(n)), "malio_map_size")
        return n.value

    def nearest_search(self, q12, k=5):
        q12 = np.ascontiguousarray(q12, np.float32)
        n = q12.shape[0]
        out = np.zeros((n, k, 12), np.float32)
        d2 = np.zeros((n, k), np.float32)
        cnt = np.zeros(n, np.int32)
        self._chk(lib().malio_nearest_search(self.h, _p(q12, Point), n, k, _p(out, Point), _p(d2, C.c_float),
                                             _p(cnt, C.c_int)), "malio_nearest_search")
        return out, d2, cnt

    def scan_set(self, pts12, pose_tables, temporal_comp):
        pts12 = np.ascontiguousarray(pts12, np.float32)
        self.N = pts12.shape[0]
        tabs = [np.ascontiguousarray(np.asarray(t, np.float64).reshape(-1, 59)) for t in pose_tables]
        ptrs = (C.POINTER(Pose) * self.L)(*[_p(t, Pose) for t in tabs])
        lens = (C.c_int * self.L)(*[t.shape[0] for t in tabs])
        tc = np.ascontiguousarray(np.asarray(temporal_comp, np.float64).reshape(-1, 59))
        tcp = _p(tc, Pose) if self.L > 1 else None
        self._keep = (tabs, tc)
        self._chk(lib().malio_scan_set(self.h, _p(pts12, Point), self.N, ptrs, lens, tcp), "malio_scan_set")

    def scan_stage(self, arr, packed=False):
        """malio_scan_stage: copy the NEXT scan's page-locked array ([n,12] points or [n,5] packed records) ahead."""
        assert arr.dtype == np.float32 and arr.flags.c_contiguous
        self._chk(lib().malio_scan_stage(self.h, C.c_void_p(arr.ctypes.data), arr.shape[0], int(bool(packed))), "malio_scan_stage")

    @staticmethod
    def pack_scan(pts12):
        """The 20-byte records of malio_scan_set_packed from 48-byte points: [n] structured as 5 x 4 bytes (x, y, z, w, ny)."""
        pts12 = np.ascontiguousarray(pts12, np.float32)
        rec = np.zeros((pts12.shape[0], 5), np.float32)
        rec[:, 0:3] = pts12[:, 0:3]
        idx = np.clip(pts12[:, 4].astype(np.int64), -0x3FFFFF, 0x3FFFFF)
        w = ((idx << 8) & 0xFFFFFFFF) | pts12[:, 8].astype(np.int64)
        rec.view(np.uint32)[:, 3] = w.astype(np.uint32)
        rec[:, 4] = pts12[:, 5]
        return rec

    def scan_set_packed_fn(self, rec5, pose_tables, temporal_comp):
        """malio_scan_set_packed with its arguments built beforehand (see scan_set_fn)."""
        rec5 = np.ascontiguousarray(rec5, np.float32) if not isinstance(rec5, np.ndarray) or rec5.dtype != np.float32 else rec5
        n = rec5.shape[0]
        tabs = [np.ascontiguousarray(np.asarray(t, np.float64).reshape(-1, 59)) for t in pose_tables]
        ptrs = (C.POINTER(Pose) * self.L)(*[_p(t, Pose) for t in tabs])
        lens = (C.c_int * self.L)(*[t.shape[0] for t in tabs])
        tc = np.ascontiguousarray(np.asarray(temporal_comp, np.float64).reshape(-1, 59))
        tcp = _p(tc, Pose) if self.L > 1 else None
        p = rec5.ctypes.data_as(C.c_void_p)
        fn = lib().malio_scan_set_packed

        def call():
            self.N = n
            self._keep = (tabs, tc, rec5)
            self._chk(fn(self.h, p, n, ptrs, lens, tcp), "malio_scan_set_packed")
        return call

    def scan_set_packed(self, rec5, pose_tables, temporal_comp):
        self.scan_set_packed_fn(rec5, pose_tables, temporal_comp)()

    def scan_upload_wait(self):
        """malio_scan_upload_wait: the page-locked cloud handed to the last scan_set may be modified / freed after this."""
        self._chk(lib().malio_scan_upload_wait(self.h), "malio_scan_upload_wait")

    def predict_chain(self, xs_flat, Ps, dts, accs, gyros, Q, want_states=True):
        """malio_predict_chain: tracks = len(xs_flat); dts/accs/gyros: per track arrays (K_t, K_t x 3, K_t x 3).
        Returns (end states flat, end P list or None, per-step states list per track)."""
        nt = len(xs_flat)
        X = (State * nt)(*[state_from_flat(x, self.L) for x in xs_flat])
        n = 17 + 6 * self.L
        P = None if Ps is None else np.ascontiguousarray(np.stack([np.asarray(p, np.float64) for p in Ps]))
        K = (C.c_int * nt)(*[len(d) for d in dts])
        dt = np.ascontiguousarray(np.concatenate([np.asarray(d, np.float64) for d in dts]))
        acc = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64).reshape(-1, 3) for a in accs]))
        gy = np.ascontiguousarray(np.concatenate([np.asarray(g, np.float64).reshape(-1, 3) for g in gyros]))
        Qc = np.ascontiguousarray(Q, np.float64)
        tot = int(dt.shape[0])
        out = (State * max(tot, 1))() if want_states else None
        self._chk(lib().malio_predict_chain(self.h, nt, X, None if P is None else _p(P, C.c_double), K, _p(dt, C.c_double),
                                            _p(acc, C.c_double), _p(gy, C.c_double), _p(Qc, C.c_double), out),
                  "malio_predict_chain")
        ends = [state_to_flat(X[t], self.L) for t in range(nt)]
        steps, o = [], 0
        for t in range(nt):
            steps.append([state_to_flat(out[o + k], self.L) for k in range(K[t])] if want_states else None)
            o += K[t]
        return ends, (None if P is None else [P[t].reshape(n, n) for t in range(nt)]), steps

    def scan_set_fn(self, pts12, pose_tables, temporal_comp):
        """malio_scan_set with every ctypes argument built beforehand: returns a zero-argument callable (bench.py times
        the C call, not the marshalling of ~30 pose-table entries in Python)."""
        pts12 = np.ascontiguousarray(pts12, np.float32)
        n = pts12.shape[0]
        tabs = [np.ascontiguousarray(np.asarray(t, np.float64).reshape(-1, 59)) for t in pose_tables]
        ptrs = (C.POINTER(Pose) * self.L)(*[_p(t, Pose) for t in tabs])
        lens = (C.c_int * self.L)(*[t.shape[0] for t in tabs])
        tc = np.ascontiguousarray(np.asarray(temporal_comp, np.float64).reshape(-1, 59))
        tcp = _p(tc, Pose) if self.L > 1 else None
        p = _p(pts12, Point)
        fn = lib().malio_scan_set

        def call():
            self.N = n
            self._keep = (tabs, tc, pts12)
            self._chk(fn(self.h, p, n, ptrs, lens, tcp), "malio_scan_set")
        return call

    def measure(self, state_flat, converge=True, want_rows=False):
        s = state_from_flat(state_flat, self.L)
        out = MeasureOut()
        if want_rows:
            hx = np.zeros((self.N, self.C), np.float64)
            hv = np.zeros(self.N, np.float64)
            Rv = np.zeros(self.N, np.float64)
            out.h_x, out.h, out.R = _p(hx, C.c_double), _p(hv, C.c_double), _p(Rv, C.c_double)
        rc = self._chk(lib().malio_measure(self.h, C.byref(s), int(bool(converge)), C.byref(out)), "malio_measure")
        Cc = self.C
        res = dict(rc=rc, valid=bool(out.valid), M=int(out.M), w_loc=float(out.w_loc),
                   HtRinvH=np.array(out.HtRinvH[:Cc * Cc]).reshape(Cc, Cc), HtRinvh=np.array(out.HtRinvh[:Cc]),
                   unit_cov_minmax=tuple(out.unit_cov_minmax), R_minmax=tuple(out.R_minmax))
        if want_rows:
            M = res["M"] if res["valid"] else 0
            res.update(h_x=hx[:M].copy(), h=hv[:M].copy(), R=Rv[:M].copy())
        return res

    def measure_fn(self, state_flat, converge=True):
        """Pre-bound call for timing loops: returns (fn, out_struct); fn() runs one malio_measure pass with no
        Python-side conversions (state struct and output struct are built once)."""
        s = state_from_flat(state_flat, self.L)
        out = MeasureOut()
        f = lib().malio_measure
        h, sp, op, cv = self.h, C.byref(s), C.byref(out), int(bool(converge))
        keep = (s, out)

        def fn():
            return f(h, sp, cv, op)
        fn._keep = keep
        return fn, out

    def update_iterated_async(self, state_flat, P):
        """malio_update_iterated_begin: returns a function that ends the update (malio_update_iterated_end) and returns the
        same dict as update_iterated; the calling thread is free in between."""
        s = state_from_flat(state_flat, self.L)
        Pw = np.ascontiguousarray(P, np.float64).copy()
        self._chk(lib().malio_update_iterated_begin(self.h, C.byref(s), _p(Pw, C.c_double)), "malio_update_iterated_begin")

        def end():
            stats = (C.c_int * 4)()
            rc = self._chk(lib().malio_update_iterated_end(self.h, C.byref(s), _p(Pw, C.c_double), stats), "malio_update_iterated_end")
            return dict(rc=rc, state=state_to_flat(s, self.L), P=Pw, passes=stats[0], searches=stats[1], M=stats[2], t=stats[3])
        return end

    def update_iterated_fn(self, state_flat, P, R=0.001):
        """Pre-bound malio_update_iterated for timing loops: returns (fn, result); fn() restores the prior (state, P) and
        runs the update - two small memcpys and the C call, no Python conversions; result() builds the dict afterwards."""
        s0 = state_from_flat(state_flat, self.L)
        s = state_from_flat(state_flat, self.L)
        P0 = np.ascontiguousarray(P, np.float64).copy()
        Pw = P0.copy()
        stats = (C.c_int * 4)()
        st = C.c_double(0)
        f = lib().malio_update_iterated
        h, sp, pp, stp, Rc = self.h, C.byref(s), _p(Pw, C.c_double), C.byref(st), C.c_double(R)
        ssz, s0p, psz, p0p, pwp = C.sizeof(s), C.addressof(s0), P0.nbytes, P0.ctypes.data, Pw.ctypes.data

        def fn():
            C.memmove(C.addressof(s), s0p, ssz)
            C.memmove(pwp, p0p, psz)
            return f(h, sp, pp, Rc, stats, stp)

        def result():
            return dict(state=state_to_flat(s, self.L), P=Pw.copy(), passes=stats[0], searches=stats[1], M=stats[2],
                        t=stats[3], solve_time=st.value)
        fn._keep = (s0, s, P0, Pw, stats, st)
        return fn, result

    def scan_get(self):
        n = self.N
        out = dict(normal_y=np.zeros(n, np.float32), nearest=np.zeros((n, 5, 12), np.float32),
                   nearest_cnt=np.zeros(n, np.int32), selected=np.zeros(n, np.uint8),
                   res_last=np.zeros(n, np.float32), world=np.zeros((n, 3), np.float32),
                   normvec=np.zeros((n, 4), np.float32))
        self._chk(lib().malio_scan_get(self.h, _p(out["normal_y"], C.c_float), _p(out["nearest"], Point),
                                       _p(out["nearest_cnt"], C.c_int), _p(out["selected"], C.c_uint8),
                                       _p(out["res_last"], C.c_float), _p(out["world"], C.c_float),
                                       _p(out["normvec"], C.c_float)), "malio_scan_get")
        return out

    def set_partition(self, rank, world, tile_m=0.0, columns=False):
        """malio_set_partition[_shape]: this handle becomes spatial shard `rank` of `world` (call before map_build)."""
        self._chk(lib().malio_set_partition_shape(self.h, int(rank), int(world), C.c_float(tile_m), TILE_COLUMNS if columns else TILE_CUBES),
                  "malio_set_partition_shape")

    def scan_owned(self):
        out = np.zeros(self.N, np.uint8)
        self._chk(lib().malio_scan_owned(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))), "malio_scan_owned")
        return out.astype(bool)

    def set_pass_hook(self, fn):
        """malio_set_pass_hook: fn(pass_number) runs on the host before every measurement pass of update_iterated."""
        proto = C.CFUNCTYPE(None, C.c_int, C.c_void_p)
        self._hook = proto(lambda k, _u: fn(k)) if fn else None
        self._chk(lib().malio_set_pass_hook(self.h, self._hook if fn else C.cast(None, proto), None), "malio_set_pass_hook")

    def update_iterated(self, state_flat, P, R=0.001):
        s = state_from_flat(state_flat, self.L)
        P = np.ascontiguousarray(P, np.float64).copy()
        stats = (C.c_int * 4)()
        st = C.c_double(0)
        self._chk(lib().malio_update_iterated(self.h, C.byref(s), _p(P, C.c_double), C.c_double(R), stats,
                                              C.byref(st)), "malio_update_iterated")
        return dict(state=state_to_flat(s, self.L), P=P, passes=stats[0], searches=stats[1], M=stats[2],
                    t=stats[3], solve_time=st.value)

    def undistort(self, pts12, lidar_beg_time, knot_times, knot_poses, ext_q, ext_t, end_q, end_t, imu_stamps,
                  cov_pointer0):
        """malio_undistort: returns (points [n,12] copy, entry point indices)."""
        pts = np.ascontiguousarray(pts12, np.float32).copy()
        kt = np.ascontiguousarray(knot_times, np.float64)
        kp = np.ascontiguousarray(knot_poses, np.float64).reshape(-1, 16)
        imu = np.ascontiguousarray(imu_stamps, np.float64)
        v = lambda a: _p(np.ascontiguousarray(a, np.float64), C.c_double)
        ent = np.zeros(max(len(imu), 1) + 4, np.int32)
        ne = C.c_int(0)
        self._chk(lib().malio_undistort(self.h, _p(pts, Point), pts.shape[0], C.c_double(lidar_beg_time),
                                        _p(kt, C.c_double), _p(kp, C.c_double), len(kt), v(ext_q), v(ext_t), v(end_q),
                                        v(end_t), _p(imu, C.c_double), len(imu), int(cov_pointer0), _p(ent, C.c_int),
                                        C.byref(ne)), "malio_undistort")
        return pts, ent[:ne.value].copy()

    def undistort_resident(self, lid, pts12, lidar_beg_time, knot_times, knot_poses, ext_q, ext_t, end_q, end_t, imu_stamps,
                           cov_pointer0):
        """malio_undistort_resident: the cloud stays in HBM; returns (entry point indices, entry points [k,12])."""
        pts = np.ascontiguousarray(pts12, np.float32)
        kt = np.ascontiguousarray(knot_times, np.float64)
        kp = np.ascontiguousarray(knot_poses, np.float64).reshape(-1, 16)
        imu = np.ascontiguousarray(imu_stamps, np.float64)
        v = lambda a: _p(np.ascontiguousarray(a, np.float64), C.c_double)
        ent = np.zeros(max(len(imu), 1) + 4, np.int32)
        epts = np.zeros((ent.shape[0], 12), np.float32)
        ne = C.c_int(0)
        self._res_n = getattr(self, "_res_n", {})
        self._res_n[int(lid)] = pts.shape[0]
        self._chk(lib().malio_undistort_resident(self.h, int(lid), _p(pts, Point), pts.shape[0], C.c_double(lidar_beg_time),
                                                 _p(kt, C.c_double), _p(kp, C.c_double), len(kt), v(ext_q), v(ext_t),
                                                 v(end_q), v(end_t), _p(imu, C.c_double), len(imu), int(cov_pointer0),
                                                 _p(ent, C.c_int), C.byref(ne), _p(epts, Point)), "malio_undistort_resident")
        return ent[:ne.value].copy(), epts[:ne.value].copy()

    def scan_set_resident(self, leaf, pose_tables, temporal_comp, normal_mode=1, want_body=True, cap=None, out=None):
        """malio_scan_set_resident: voxel filter + scan upload from the resident clouds. Returns feats_down_body
        (a view of `out` when the caller supplies the buffer, e.g. a PinnedArray's array)."""
        L = self.L
        arrs = [np.ascontiguousarray(t, np.float64).reshape(-1, 59) for t in pose_tables]
        ptrs = (C.POINTER(Pose) * L)(*[a.ctypes.data_as(C.POINTER(Pose)) for a in arrs])
        lens = (C.c_int * L)(*[a.shape[0] for a in arrs])
        tc = np.ascontiguousarray(temporal_comp, np.float64).reshape(-1, 59) if L > 1 else None
        cap = int(cap if cap is not None else sum(getattr(self, "_res_n", {}).values()))
        own = out is None
        if own:
            out = np.zeros((cap if want_body else 1, 12), np.float32)
        else:
            assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape[1] == 12
            cap = min(cap, out.shape[0])
        n = C.c_int(0)
        self._chk(lib().malio_scan_set_resident(self.h, C.c_float(leaf), int(normal_mode), ptrs, lens,
                                                tc.ctypes.data_as(C.POINTER(Pose)) if tc is not None else None,
                                                _p(out, Point) if want_body else None, cap if want_body else 0,
                                                C.byref(n)), "malio_scan_set_resident")
        self.N = n.value
        self._res_n = {}
        if not want_body:
            return None
        return out[:n.value].copy() if own else out[:n.value]

    def scan_order(self, mode):
        """malio_scan_order: 0 auto (host scans sorted, resident scans as they are), 1 always sort, 2 never sort."""
        self._chk(lib().malio_scan_order(self.h, int(mode)), "malio_scan_order")

    def measure_node_fn(self, xchg, state_flat, converge=True):
        """Pre-bound malio_measure_node (one pass over a scan / map sharded across ranks): fn() -> rc, out struct."""
        s = state_from_flat(state_flat, self.L)
        out = MeasureOut()
        f, h, xh, sp, op, cv = lib().malio_measure_node, self.h, xchg.h, C.byref(s), C.byref(out), int(bool(converge))

        def fn():
            return f(h, xh, sp, cv, op, None)
        fn._keep = (s, out, xchg)
        return fn, out

    def update_iterated_node(self, xchg, state_flat, P, R=0.001):
        """malio_update_iterated_node: the iterated update over a scan sharded across the ranks of one node."""
        n = 17 + 6 * self.L
        s = state_from_flat(state_flat, self.L)
        P = np.array(P, np.float64, order="C").reshape(n, n)
        stats = (C.c_int * 4)()
        st = C.c_double(0)
        rc = self._chk(lib().malio_update_iterated_node(self.h, xchg.h, C.byref(s), _p(P, C.c_double), C.c_double(R), stats,
                                                        C.byref(st)), "malio_update_iterated_node")
        return dict(rc=rc, state=state_to_flat(s, self.L), P=P, passes=int(stats[0]), searches=int(stats[1]),
                    M=int(stats[2]), solve_time=st.value)

    def node_stats(self):
        """(passes of malio_measure_node that needed one exchange, passes that needed two) so far."""
        st = (C.c_int * 2)()
        self._chk(lib().malio_node_stats(self.h, st), "malio_node_stats")
        return int(st[0]), int(st[1])

    def result_buffer(self):
        """malio_result_buffer: (float64 NumPy view of the pinned host buffer, device alias as int)."""
        hp, dp, n = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.c_int(0)
        self._chk(lib().malio_result_buffer(self.h, C.byref(hp), C.byref(dp), C.byref(n)), "malio_result_buffer")
        view = np.ctypeslib.as_array(hp, shape=(n.value,))
        return view, C.cast(dp, C.c_void_p).value

    # ---- multi-GPU staging (device pointers are plain ints, e.g. torch.Tensor.data_ptr()) ----
    def sums_len(self):
        return lib().malio_sums_len(self.h)

    def stage1(self, state_flat, converge, d_minmax_ptr):
        s = state_from_flat(state_flat, self.L)
        self._chk(lib().malio_measure_stage1(self.h, C.byref(s), int(bool(converge)), C.c_void_p(d_minmax_ptr)),
                  "malio_measure_stage1")

    def stage2(self, d_minmax_ptr, d_sums_ptr):
        self._chk(lib().malio_measure_stage2(self.h, C.c_void_p(d_minmax_ptr), C.c_void_p(d_sums_ptr)),
                  "malio_measure_stage2")

    def finish(self, sums_host, minmax_host):
        sums_host = np.ascontiguousarray(sums_host, np.float64)
        minmax_host = np.ascontiguousarray(minmax_host, np.float64)
        assert minmax_host.shape[0] >= 8, "the extrema buffer is MALIO_MINMAX_LEN = 8 doubles"
        out = MeasureOut()
        rc = self._chk(lib().malio_measure_finish(self.h, _p(sums_host, C.c_double), _p(minmax_host, C.c_double),
                                                  C.byref(out)), "malio_measure_finish")
        Cc = self.C
        return dict(rc=rc, valid=bool(out.valid), M=int(out.M), w_loc=float(out.w_loc),
                    HtRinvH=np.array(out.HtRinvH[:Cc * Cc]).reshape(Cc, Cc), HtRinvh=np.array(out.HtRinvh[:Cc]))


def ieskf_step(L, max_iteration, i, x_flat, xprop_flat, P_prop, HtRinvH, HtRinvh, t, limit=0.0):
    """malio_ieskf_step (pure host, no GPU). Returns (x_new_flat, t, converge, done, P_out)."""
    n = 17 + 6 * L
    x = state_from_flat(x_flat, L)
    xp = state_from_flat(xprop_flat, L)
    P_prop = np.ascontiguousarray(P_prop, np.float64)
    H = np.ascontiguousarray(HtRinvH, np.float64)
    hv = np.ascontiguousarray(HtRinvh, np.float64)
    P_out = np.zeros((n, n), np.float64)
    t_io, conv, done = C.c_int(int(t)), C.c_int(0), C.c_int(0)
    rc = lib().malio_ieskf_step(int(L), int(max_iteration), C.c_double(limit), int(i), C.byref(x), C.byref(xp), _p(P_prop, C.c_double),
                                _p(H, C.c_double), _p(hv, C.c_double), C.byref(t_io), C.byref(conv), C.byref(done),
                                _p(P_out, C.c_double))
    if rc != OK:
        raise MalioError(f"malio_ieskf_step rc={rc}")
    return state_to_flat(x, L), t_io.value, bool(conv.value), bool(done.value), P_out


class NodeExchange:
    """malio_xchg_*: all-gather of one row of doubles between the ranks of one node through shared memory."""

    def __init__(self, name, rank, world, row_doubles, create, timeout_s=60.0):
        self.h = C.c_void_p()
        self.world, self.row, self.timeout = int(world), int(row_doubles), float(timeout_s)
        rc = lib().malio_xchg_create(name.encode(), int(rank), int(world), int(row_doubles), int(bool(create)), C.byref(self.h))
        if rc != OK:
            raise MalioError(f"malio_xchg_create({name}) rc={rc}")
        self.out = np.zeros((self.world, self.row), np.float64)
        self._fn = lib().malio_xchg_all_gather
        self._out_p = _p(self.out, C.c_double)
        self._to = C.c_double(self.timeout)

    def all_gather(self, row):
        """row: contiguous float64 array of row_doubles entries. Returns the [world, row] view (overwritten by the
        next call)."""
        rc = self._fn(self.h, _p(row, C.c_double), self._out_p, self._to)
        if rc != OK:
            raise MalioError(f"malio_xchg_all_gather rc={rc} (a rank is missing?)")
        return self.out

    def unlink(self):
        """Creator, after every rank opened the segment: drop the name so that nothing outlives the job."""
        lib().malio_xchg_unlink(self.h)

    def close(self):
        if self.h:
            lib().malio_xchg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RcclExchange(NodeExchange):
    """malio_xchg_create_rccl: the same exchange over RCCL (one process or thread per GPU). unique_id: bytes from
    rccl_unique_id() on one rank, distributed by the launcher (e.g. torch.distributed.broadcast_object_list)."""

    def __init__(self, unique_id, rank, world, row_doubles, device):
        self.h = C.c_void_p()
        self.world, self.row, self.timeout = int(world), int(row_doubles), 60.0
        buf = C.create_string_buffer(bytes(unique_id), 128)
        rc = lib().malio_xchg_create_rccl(buf, int(rank), int(world), int(row_doubles), int(device), C.byref(self.h))
        if rc != OK:
            raise MalioError(f"malio_xchg_create_rccl rc={rc}")
        self.out = np.zeros((self.world, self.row), np.float64)
        self._fn = lib().malio_xchg_all_gather
        self._out_p = _p(self.out, C.c_double)
        self._to = C.c_double(self.timeout)

    def unlink(self):
        pass


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().malio_rccl_unique_id(buf)
    if rc != OK:
        raise MalioError(f"malio_rccl_unique_id rc={rc}")
    return bytes(buf.raw)


def part_owner(xyz, world, tile_m=0.0, columns=False):
    """malio_part_owner[_shape] for an [n,3] float32 array: the shard that serves a world point."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.zeros(xyz.shape[0], np.int32)
    rc = lib().malio_part_owner_shape(_p(xyz, C.c_float), xyz.shape[0], int(world), C.c_float(tile_m),
                                      TILE_COLUMNS if columns else TILE_CUBES, _p(out, C.c_int))
    assert rc == OK
    return out


def part_stores(xyz, rank, world, tile_m=0.0, filter_size_map=0.5, columns=False):
    """malio_part_stores[_shape] for an [n,3] float32 array: whether shard `rank` keeps a map point (own tiles + halo)."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.zeros(xyz.shape[0], np.uint8)
    rc = lib().malio_part_stores_shape(_p(xyz, C.c_float), xyz.shape[0], int(rank), int(world), C.c_float(tile_m),
                                       TILE_COLUMNS if columns else TILE_CUBES, C.c_float(filter_size_map), _p(out, C.c_uint8))
    assert rc == OK
    return out.astype(bool)


class Node:
    """malio_node_*: several GPUs (or several shards on one GPU: devices=[0, 0, ...]) behind one handle."""

    def __init__(self, params: dict, devices, partition=PART_SCAN, exchange=XCHG_HOST, tile_m=0.0, cell_size=0.0):
        self.L = int(params["lid_num"])
        self.C = 6 * (1 + self.L)
        self.n = 17 + 6 * self.L
        self.params = dict(params)
        self._prm = make_params(params, cell_size)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self.G = len(devices)
        self.h = C.c_void_p()
        rc = lib().malio_node_create(C.byref(self._prm), self.G, devs, int(partition), int(exchange), C.c_float(tile_m),
                                     C.byref(self.h))
        if rc != OK:
            self.h = None
            raise MalioError(f"malio_node_create rc={rc}")
        self.N = 0

    def _chk(self, rc, what):
        if rc < 0:
            lib().malio_node_last_error.restype = C.c_char_p
            lib().malio_node_last_error.argtypes = [C.c_void_p]
            raise MalioError(f"{what} rc={rc}: {lib().malio_node_last_error(self.h).decode()}")
        return rc

    def close(self):
        if self.h:
            lib().malio_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def map_build(self, pts12):
        pts12 = np.ascontiguousarray(pts12, np.float32)
        self._chk(lib().malio_node_map_build(self.h, _p(pts12, Point), pts12.shape[0]), "malio_node_map_build")

    def map_sizes(self):
        out = (C.c_int * self.G)()
        self._chk(lib().malio_node_map_size(self.h, out), "malio_node_map_size")
        return list(out)

    def map_delete_boxes(self, boxes6):
        boxes6 = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
        out = (C.c_int * self.G)()
        self._chk(lib().malio_node_map_delete_boxes(self.h, boxes6.ctypes.data_as(C.c_void_p), boxes6.shape[0], out),
                  "malio_node_map_delete_boxes")
        return list(out)

    def map_add(self, pts12, downsample_on=True):
        pts12 = np.ascontiguousarray(pts12, np.float32).reshape(-1, 12)
        out = (C.c_int * self.G)()
        self._chk(lib().malio_node_map_add(self.h, _p(pts12, Point), pts12.shape[0], int(bool(downsample_on)), out),
                  "malio_node_map_add")
        return list(out)

    def scan_set(self, pts12, pose_tables, temporal_comp):
        pts12 = np.ascontiguousarray(pts12, np.float32)
        self.N = pts12.shape[0]
        self._scan_keep = _scan_tables(self.L, pose_tables, temporal_comp)
        ptrs, lens, tcp = self._scan_keep[:3]
        self._chk(lib().malio_node_scan_set(self.h, _p(pts12, Point), self.N, ptrs, lens, tcp), "malio_node_scan_set")

    def measure(self, state_flat, converge=True):
        s = state_from_flat(state_flat, self.L)
        out = MeasureOut()
        rc = self._chk(lib().malio_node_measure(self.h, C.byref(s), int(bool(converge)), C.byref(out)), "malio_node_measure")
        Cc = self.C
        return dict(rc=rc, valid=bool(out.valid), M=int(out.M), w_loc=float(out.w_loc),
                    HtRinvH=np.array(out.HtRinvH[:Cc * Cc]).reshape(Cc, Cc), HtRinvh=np.array(out.HtRinvh[:Cc]),
                    unit_cov_minmax=tuple(out.unit_cov_minmax), R_minmax=tuple(out.R_minmax))

    def measure_fn(self, state_flat, converge=True):
        """Pre-bound pass for timing loops: fn() -> rc."""
        s = state_from_flat(state_flat, self.L)
        out = MeasureOut()
        f, h, sp, op, cv = lib().malio_node_measure, self.h, C.byref(s), C.byref(out), int(bool(converge))

        def fn():
            return f(h, sp, cv, op)
        fn._keep = (s, out)
        return fn, out

    def update_iterated(self, state_flat, P, R=0.001):
        s = state_from_flat(state_flat, self.L)
        P = np.ascontiguousarray(P, np.float64).copy()
        stats = (C.c_int * 4)()
        st = C.c_double(0)
        rc = self._chk(lib().malio_node_update_iterated(self.h, C.byref(s), _p(P, C.c_double), C.c_double(R), stats,
                                                        C.byref(st)), "malio_node_update_iterated")
        return dict(rc=rc, state=state_to_flat(s, self.L), P=P, passes=stats[0], searches=stats[1], M=stats[2], t=stats[3],
                    solve_time=st.value)

    def scan_get(self):
        n = self.N
        out = dict(normal_y=np.zeros(n, np.float32), nearest=np.zeros((n, 5, 12), np.float32),
                   nearest_cnt=np.zeros(n, np.int32), selected=np.zeros(n, np.uint8),
                   res_last=np.zeros(n, np.float32), world=np.zeros((n, 3), np.float32),
                   normvec=np.zeros((n, 4), np.float32))
        self._chk(lib().malio_node_scan_get(self.h, _p(out["normal_y"], C.c_float), _p(out["nearest"], Point),
                                            _p(out["nearest_cnt"], C.c_int), _p(out["selected"], C.c_uint8),
                                            _p(out["res_last"], C.c_float), _p(out["world"], C.c_float),
                                            _p(out["normvec"], C.c_float)), "malio_node_scan_get")
        return out

    def map_incremental(self, state_flat, flg_EKF_inited=True, world_normal_y=None):
        """malio_node_map_incremental. Returns (|PointToAdd|, |PointNoNeedDownsample|, added on GPU 0)."""
        s = state_from_flat(state_flat, self.L)
        wny = None if world_normal_y is None else np.ascontiguousarray(world_normal_y, np.float32)
        cnt = (C.c_int * 3)()
        self._chk(lib().malio_node_map_incremental(self.h, C.byref(s), int(bool(flg_EKF_inited)),
                                                   None if wny is None else _p(wny, C.c_float), cnt), "malio_node_map_incremental")
        return int(cnt[0]), int(cnt[1]), int(cnt[2])

    def undistort_resident(self, lid, pts12, lidar_beg_time, knot_times, knot_poses, ext_q, ext_t, end_q, end_t, imu_stamps,
                           cov_pointer0):
        """malio_node_undistort_resident: LiDAR `lid`'s cloud stays on GPU lid % n; returns the entry point indices."""
        pts = np.ascontiguousarray(pts12, np.float32)
        kt = np.ascontiguousarray(knot_times, np.float64)
        kp = np.ascontiguousarray(knot_poses, np.float64).reshape(-1, 16)
        imu = np.ascontiguousarray(imu_stamps, np.float64)
        v = lambda a: _p(np.ascontiguousarray(a, np.float64), C.c_double)
        ent = np.zeros(max(len(imu), 1) + 4, np.int32)
        ne = C.c_int(0)
        self._res_n = getattr(self, "_res_n", {})
        self._res_n[int(lid)] = pts.shape[0]
        self._chk(lib().malio_node_undistort_resident(self.h, int(lid), _p(pts, Point), pts.shape[0], C.c_double(lidar_beg_time),
                                                      _p(kt, C.c_double), _p(kp, C.c_double), len(kt), v(ext_q), v(ext_t),
                                                      v(end_q), v(end_t), _p(imu, C.c_double), len(imu), int(cov_pointer0),
                                                      _p(ent, C.c_int), C.byref(ne), None), "malio_node_undistort_resident")
        return ent[:ne.value].copy()

    def scan_set_resident(self, leaf, pose_tables, temporal_comp, normal_mode=1):
        """malio_node_scan_set_resident: returns feats_down_body."""
        self._scan_keep = _scan_tables(self.L, pose_tables, temporal_comp)
        ptrs, lens, tcp = self._scan_keep[:3]
        cap = int(sum(getattr(self, "_res_n", {}).values()))
        out = np.zeros((max(cap, 1), 12), np.float32)
        n = C.c_int(0)
        self._chk(lib().malio_node_scan_set_resident(self.h, C.c_float(leaf), int(normal_mode), ptrs, lens, tcp, _p(out, Point), cap,
                                                     C.byref(n)), "malio_node_scan_set_resident")
        self.N = n.value
        self._res_n = {}
        return out[:n.value].copy()

    def nearest_search(self, q12, k=5):
        q12 = np.ascontiguousarray(q12, np.float32)
        n = q12.shape[0]
        out = np.zeros((n, k, 12), np.float32)
        d2 = np.zeros((n, k), np.float32)
        cnt = np.zeros(n, np.int32)
        self._chk(lib().malio_node_nearest_search(self.h, _p(q12, Point), n, k, _p(out, Point), _p(d2, C.c_float),
                                                  _p(cnt, C.c_int)), "malio_node_nearest_search")
        return out, d2, cnt

    def map_get(self, rank):
        """The map points GPU `rank` holds (malio_map_get on its handle)."""
        hh = C.c_void_p()
        self._chk(lib().malio_node_handle(self.h, int(rank), C.byref(hh)), "malio_node_handle")
        n = C.c_int(0)
        lib().malio_map_get(hh, None, 0, C.byref(n))
        out = np.zeros((max(n.value, 1), 12), np.float32)
        lib().malio_map_get(hh, _p(out, Point), n.value, C.byref(n))
        return out[:n.value]

    def set_option(self, name, value):
        f = lib().malio_node_set_option
        f.argtypes = [C.c_void_p, C.c_int, C.c_double]
        self._chk(f(self.h, int(OPT.get(name, name)), float(value)), "malio_node_set_option(%s)" % name)
        return self

    def set_option_rank(self, rank, name, value):
        """one shard's handle only (tests: a stall on ONE shard)"""
        hh = C.c_void_p()
        self._chk(lib().malio_node_handle(self.h, int(rank), C.byref(hh)), "malio_node_handle")
        f = lib().malio_set_option
        f.argtypes = [C.c_void_p, C.c_int, C.c_double]
        self._chk(f(hh, int(OPT.get(name, name)), float(value)), "malio_set_option(%s)" % name)
        return self

    def get_option_rank(self, rank, name):
        hh = C.c_void_p()
        self._chk(lib().malio_node_handle(self.h, int(rank), C.byref(hh)), "malio_node_handle")
        f = lib().malio_get_option
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        v = C.c_double(0)
        self._chk(f(hh, int(OPT.get(name, name)), C.byref(v)), "malio_get_option(%s)" % name)
        return v.value

    def exchange_stats(self):
        st = (C.c_int * 2)()
        lib().malio_node_exchange_stats(self.h, st)
        return int(st[0]), int(st[1])

    def update_stats(self):
        """Summed over the shards: updates through the gated chain, of those handed back to the pass-by-pass loop, gate time-outs."""
        st = (C.c_int * 4)()
        self._chk(lib().malio_node_update_stats(self.h, st), "malio_node_update_stats")
        return dict(gated_runs=int(st[0]), gated_redone=int(st[1]), gate_timeouts=int(st[2]))

    def set_pass_hook(self, fn):
        proto = C.CFUNCTYPE(None, C.c_int, C.c_void_p)
        self._hook = proto(lambda k, _u: fn(k)) if fn else None
        self._chk(lib().malio_node_set_pass_hook(self.h, self._hook if fn else C.cast(None, proto), None), "malio_node_set_pass_hook")


class PinnedArray:
    """NumPy view of a page-locked host buffer (malio_host_alloc); keep the object alive while the view is used."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = C.c_void_p()
        rc = lib().malio_host_alloc(C.c_size_t(max(nbytes, 1)), C.byref(self.ptr))
        if rc != OK:
            raise MalioError(f"malio_host_alloc rc={rc}")
        buf = (C.c_char * max(nbytes, 1)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def __del__(self):
        try:
            if self.ptr:
                lib().malio_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def predict(L, x_flat, P, dt, Q, acc, gyro):
    """malio_predict (pure host, no GPU): one esekf::predict step. Returns (x_new_flat, P_new)."""
    x = state_from_flat(x_flat, L)
    P = None if P is None else np.array(P, np.float64, order="C")
    Q = np.ascontiguousarray(Q, np.float64)
    acc = np.ascontiguousarray(acc, np.float64)
    gyro = np.ascontiguousarray(gyro, np.float64)
    rc = lib().malio_predict(int(L), C.byref(x), None if P is None else _p(P, C.c_double), C.c_double(dt),
                             _p(Q, C.c_double), _p(acc, C.c_double), _p(gyro, C.c_double))
    if rc != OK:
        raise MalioError(f"malio_predict rc={rc}")
    return state_to_flat(x, L), P


def spline_feed(traj8, cap=4096):
    """malio_spline_feed (pure host): control-point times [K] and poses [K,4,4]."""
    traj8 = np.ascontiguousarray(traj8, np.float64)
    t = np.zeros(cap, np.float64)
    T = np.zeros((cap, 16), np.float64)
    n = C.c_int(0)
    rc = lib().malio_spline_feed(_p(traj8, C.c_double), traj8.shape[0], _p(t, C.c_double), _p(T, C.c_double), cap,
                                 C.byref(n))
    if rc != OK:
        raise MalioError(f"malio_spline_feed rc={rc}")
    return t[:n.value].copy(), T[:n.value].reshape(-1, 4, 4).copy()


def spline_get_pose(times, poses, ts):
    times = np.ascontiguousarray(times, np.float64)
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
    q = np.zeros(4, np.float64)
    p = np.zeros(3, np.float64)
    ok = lib().malio_spline_get_pose(_p(times, C.c_double), _p(poses, C.c_double), len(times), C.c_double(ts),
                                     _p(q, C.c_double), _p(p, C.c_double))
    return ok == 1, q, p


def compound(pose1_59, pose2_59, inverse=False, alias=False):
    """malio_compound_pose_cov / malio_compound_inv_pose_cov on flat 59-double poses (pure host)."""
    p1 = np.ascontiguousarray(pose1_59, np.float64).copy()
    p2 = np.ascontiguousarray(pose2_59, np.float64).copy()
    out = p2 if alias else np.zeros(59, np.float64)
    f = lib().malio_compound_inv_pose_cov if inverse else lib().malio_compound_pose_cov
    rc = f(_p(p1, Pose), _p(p2, Pose), _p(out, Pose))
    if rc != OK:
        raise MalioError(f"compound rc={rc}")
    return out


def eval_point_uncertainty(p12, pose59):
    p = np.ascontiguousarray(p12, np.float32)
    ps = np.ascontiguousarray(pose59, np.float64)
    cov = np.zeros(9, np.float64)
    lib().malio_eval_point_uncertainty(_p(p, Point), _p(ps, Pose), _p(cov, C.c_double))
    return cov.reshape(3, 3)
