"""Multi-GPU host logic for the measurement update: scan points sharded across ranks,
map replicated, two tiny collectives per pass - MAX over [max_unit_cov, -min_unit_cov, max_R, -min_R]
between the search/plane stage and the row stage (the FIC weights of laserMapping.cpp:651-656,716-721 are
scan-global), then SUM over the per-LiDAR 12x12 normal-equation blocks. One process per GPU,
torch.distributed ("nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

A `backend` supplies the two local stages:
    stage1(state, converge) -> torch tensor [max_u, -min_u, max_R, -min_R, M_local]   (on the backend's device)
    stage2(minmax_tensor)   -> torch tensor [L * 97] local sums
HipBackend drives libmalio_hip through the C ABI; the CPU tests plug in an oracle-backed stand-in.
"""
import numpy as np
import torch
import torch.distributed as dist

NSUM = 97  # 78 (12x12 upper) + 12 (rhs) + 6 (c^2 n n^T) + 1 (count), see csrc/measure.hip


def assemble(sums, L, prm):
    """Host finish on the reduced sums == finish_host() in csrc/measure.hip: C x C normal equations from the
    per-LiDAR blocks, localization weight (laserMapping.cpp:745-759). Returns dict(valid, M, HtRinvH, HtRinvh, w_loc)."""
    sums = np.asarray(sums, np.float64).reshape(L, NSUM)
    C = 6 * (1 + L)
    H = np.zeros((C, C))
    h = np.zeros(C)
    NtN = np.zeros((3, 3))
    M = 0.0
    iu = np.triu_indices(12)
    for l in range(L):
        gi = np.array(list(range(6)) + [6 + 3 * l + k for k in range(3)] + [6 + 3 * (L + l) + k for k in range(3)])
        blk = np.zeros((12, 12))
        blk[iu] = sums[l, :78]
        blk = blk + np.triu(blk, 1).T
        H[np.ix_(gi, gi)] += blk
        h[gi] += sums[l, 78:90]
        s = sums[l, 90:96]
        NtN += np.array([[s[0], s[3], s[4]], [s[3], s[1], s[5]], [s[4], s[5], s[2]]])
        M += sums[l, 96]
    M = int(round(M))
    if M < 1:
        return dict(valid=False, M=0, HtRinvH=H, HtRinvh=h, w_loc=0.0)
    ev = np.linalg.eigvalsh(NtN)
    w = np.sqrt(max(ev[0], 0.0)) / np.sqrt(ev[2])
    if w > prm["localize_thresh_max"]:
        w = prm["localize_cov_max"]
    elif w < prm["localize_thresh_min"]:
        w = prm["localize_cov_min"]
    else:
        w = ((prm["localize_cov_max"] - prm["localize_cov_min"]) * (w - prm["localize_thresh_min"]) /
             (prm["localize_thresh_max"] - prm["localize_thresh_min"]) + prm["localize_cov_min"])
    return dict(valid=True, M=M, HtRinvH=H * w * w, HtRinvh=h * w * w, w_loc=float(w))


class HipBackend:
    """libmalio_hip on the current CUDA/HIP device; buffers are torch tensors so RCCL can reduce them in place."""

    def __init__(self, engine):
        self.eng = engine
        self.L = engine.L
        self.params = engine.params
        self.d_mm = torch.zeros(8, dtype=torch.float64, device="cuda")
        self.d_sums = torch.zeros(engine.sums_len(), dtype=torch.float64, device="cuda")
        engine.set_stream(torch.cuda.current_stream().cuda_stream)

    def stage1(self, state, converge):
        self.eng.stage1(state, converge, self.d_mm.data_ptr())
        return self.d_mm[:5]

    def stage2(self, mm):
        self.eng.stage2(self.d_mm.data_ptr(), self.d_sums.data_ptr())
        return self.d_sums

    def pass_fn(self, state, converge, group=None):
        """Pre-bound sharded pass for loops that repeat the same (state, converge): no Python-side conversions,
        one D2H copy of [sums | extrema] into pinned memory, finish in C (malio_measure_finish). Returns
        (fn, out_struct); fn() -> rc like malio_measure. Same sequence as sharded_measure()."""
        import ctypes as C
        from . import capi
        eng = self.eng
        ns = eng.sums_len()
        buf = torch.zeros(ns + 8, dtype=torch.float64, device="cuda")          # [sums | mm(5) + pad]
        host = torch.zeros(ns + 8, dtype=torch.float64).pin_memory()
        sums, mm4 = buf[:ns], buf[ns:ns + 4]
        p_sums, p_mm = buf.data_ptr(), buf.data_ptr() + 8 * ns
        s = capi.state_from_flat(state, eng.L)
        out = capi.MeasureOut()
        lib = capi.lib()
        f1, f2, f3 = lib.malio_measure_stage1, lib.malio_measure_stage2, lib.malio_measure_finish
        h, sp, op, cv = eng.h, C.byref(s), C.byref(out), int(bool(converge))
        vp_mm, vp_sums = C.c_void_p(p_mm), C.c_void_p(p_sums)
        hp_sums = C.cast(host.data_ptr(), C.POINTER(C.c_double))
        hp_mm = C.cast(host.data_ptr() + 8 * ns, C.POINTER(C.c_double))
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        stream = torch.cuda.current_stream()

        def fn():
            rc = f1(h, sp, cv, vp_mm)
            if rc < 0:
                return rc
            if multi:
                dist.all_reduce(mm4, op=dist.ReduceOp.MAX, group=group)
            rc = f2(h, vp_mm, vp_sums)
            if rc < 0:
                return rc
            if multi:
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
            host.copy_(buf, non_blocking=True)
            stream.synchronize()
            return f3(h, hp_sums, hp_mm, op)
        fn._keep = (s, out, buf, host)
        return fn, out


def sharded_measure(backend, state, converge, group=None):
    """One measurement pass over a scan sharded across the ranks of `group`."""
    mm = backend.stage1(state, converge)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(mm[:4], op=dist.ReduceOp.MAX, group=group)
    sums = backend.stage2(mm)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return assemble(sums.detach().cpu().numpy(), backend.L, backend.params)


def sharded_update_iterated(backend, state0, P0, group=None):
    """esekfom.hpp:495-721 with the sharded pass as h_dyn_share; the n x n algebra runs (redundantly, on the
    same reduced sums, hence bit-identically) on every rank through malio_ieskf_step."""
    from . import capi
    L = backend.L
    max_it = int(backend.params["max_iteration"])
    x = np.array(state0, np.float64)
    x_prop = x.copy()
    P = np.array(P0, np.float64)
    converge, t, passes, searches, M = True, 0, 0, 0, 0
    n = 17 + 6 * L
    for i in range(-1, max_it):
        searches += int(converge)
        out = sharded_measure(backend, x, converge, group)
        passes += 1
        if not out["valid"]:
            continue
        M = out["M"]
        if M < n:
            raise NotImplementedError("M < n fallback (esekfom.hpp:574-582) is a single-GPU path")
        x, t, converge, done, P_out = capi.ieskf_step(L, max_it, i, x, x_prop, P0, out["HtRinvH"], out["HtRinvh"], t)
        if done:
            P = P_out
            break
    return dict(state=x, P=P, passes=passes, searches=searches, M=M, t=t)
