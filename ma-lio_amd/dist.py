"""Multi-GPU host logic for the measurement update: scan points sharded across ranks,
map replicated, two tiny exchanges per pass - MAX over [max_unit_cov, -min_unit_cov, max_R, -min_R]
between the search/plane stage and the row stage (the FIC weights of laserMapping.cpp:651-656,716-721 are
scan-global), then SUM over the per-LiDAR 12x12 normal-equation blocks. One process per GPU,
torch.distributed ("nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

A `backend` supplies the two local stages:
    stage1(state, converge) -> torch tensor [max_u, -min_u, max_R, -min_R, M_local]   (on the backend's device)
    stage2(minmax_tensor)   -> torch tensor [L * 97] local sums
HipBackend drives libmalio_hip through the C ABI; the CPU tests plug in an oracle-backed stand-in.
"""
import itertools
import os
import socket

import numpy as np
import torch
import torch.distributed as dist

_xchg_ids = itertools.count()


def _single_node(group=None):
    """True when every rank of `group` runs on this host."""
    lw, w = os.environ.get("LOCAL_WORLD_SIZE"), os.environ.get("WORLD_SIZE")
    if group is None and lw and w:                                              # the launcher told us
        return int(lw) == int(w)
    names = [None] * dist.get_world_size(group)
    dist.all_gather_object(names, socket.gethostname(), group=group)
    return len(set(names)) == 1


def _agree(ok, group=None):
    """Logical AND of `ok` over the ranks (a device all-reduce: works with every backend the ranks already use)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def exchange_mode(group=None):
    """How the per-pass rows travel: MALIO_EXCHANGE = shm | collective | auto (default: shared memory when all ranks
    share a node - the contract of bench.py - else the collective of the process group)."""
    mode = os.environ.get("MALIO_EXCHANGE", "auto")
    if mode not in ("shm", "collective", "auto"):
        raise ValueError("MALIO_EXCHANGE must be shm, collective or auto")
    if mode == "auto":
        mode = "shm" if _single_node(group) else "collective"
    return mode

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

    def pass_fn(self, state, converge, group=None, speculate=True):
        """Pre-bound sharded pass for loops that repeat the same (state, converge): no Python-side conversions,
        pinned staging, finish in C (malio_measure_finish). Returns (fn, out_struct); fn() -> rc like malio_measure.

        Exchanges per pass. The plain sequence (sharded_measure) needs two dependent ones: MAX of the four
        extrema before the rows can be weighted, then SUM of the normal equations. With `speculate`, a pass first
        weights its rows with the extrema of the PREVIOUS pass and sends [local sums | local extrema] in ONE
        all-gather; every rank then forms the true extrema from the gathered rows. If they equal the guess (the
        usual case from the second pass on: the extreme points of a scan rarely change between passes) the sums are
        the ones the reference would form and the pass is done after a single collective; otherwise stage 2 is run
        again with the true extrema and a second all-gather follows - the result is exact either way. The ranks add
        the gathered rows in rank order, so every rank holds the same bits.

        How the rows travel (exchange_mode): between the ranks of one node through shared memory
        (capi.NodeExchange / malio_xchg_*) - the 2.4 KB are needed on the HOST, where the filter algebra runs, so a
        device collective would only add a GPU round trip to a latency-bound message; across nodes, or with
        MALIO_EXCHANGE=collective, through all_gather_into_tensor of the process group (RCCL on the GPU box)."""
        import ctypes as C
        from . import capi
        eng = self.eng
        ns = eng.sums_len()
        row = ns + 8
        buf = torch.zeros(row, dtype=torch.float64, device="cuda")          # [sums | extrema words (8)]
        host = torch.zeros(row, dtype=torch.float64).pin_memory()
        hostn = host.numpy()
        sums, mm4 = buf[:ns], buf[ns:ns + 4]
        p_sums, p_mm = buf.data_ptr(), buf.data_ptr() + 8 * ns
        s = capi.state_from_flat(state, eng.L)
        out = capi.MeasureOut()
        lib = capi.lib()
        f1, f2, f3 = lib.malio_measure_stage1, lib.malio_measure_stage2, lib.malio_measure_finish
        f2e = lib.malio_measure_stage2_emit
        h, sp, op, cv = eng.h, C.byref(s), C.byref(out), int(bool(converge))
        vp_mm, vp_sums = C.c_void_p(p_mm), C.c_void_p(p_sums)
        hp_sums = C.cast(host.data_ptr(), C.POINTER(C.c_double))
        hp_mm = C.cast(host.data_ptr() + 8 * ns, C.POINTER(C.c_double))
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        stream = torch.cuda.current_stream()
        if not multi:
            fm = lib.malio_measure                                              # one rank: nothing to exchange

            def fn1():
                return fm(h, sp, cv, op)
            fn1._keep = (s, out)
            return fn1, out

        W, rank = dist.get_world_size(group), dist.get_rank(group)
        mode = exchange_mode(group)
        mmg = torch.zeros(8, dtype=torch.float64, device="cuda")               # the extrema stage 2 is run with
        vp_mmg = C.c_void_p(mmg.data_ptr())
        e_pin = torch.zeros(8, dtype=torch.float64).pin_memory()
        e_np = e_pin.numpy()
        st = {"guess": None, "hits": 0, "misses": 0, "exchange": mode}
        self.spec_stats = st
        xchg = None
        if mode == "shm":
            # every rank derives the same segment name (the calls are collective, so the counters agree); rank 0 creates
            # and zeroes it, the first agreement doubles as "it exists before anybody else opens it"
            # the name carries a random token rank 0 draws and broadcasts: two independent jobs on one node (same W, no
            # TORCHELASTIC_RUN_ID / MASTER_PORT to tell them apart) can never meet in - or unlink - each other's segment
            import uuid
            box = [uuid.uuid4().hex[:16] if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            name = "/malio_%s_%d_%d" % (box[0], W, next(_xchg_ids))
            ok = True
            if rank == 0:
                try:                                                            # no /dev/shm, no space: use the group
                    xchg = capi.NodeExchange(name, 0, W, row, create=True)
                except capi.MalioError:
                    ok = False
            ok = _agree(ok, group)
            if ok and rank != 0:
                try:
                    xchg = capi.NodeExchange(name, rank, W, row, create=False)
                except capi.MalioError:
                    ok = False
            all_open = _agree(ok, group)
            if rank == 0 and xchg is not None:
                xchg.unlink()                                                   # mapped everywhere (or abandoned)
            if not all_open:
                if xchg is not None:
                    xchg.close()
                xchg = None
                mode = st["exchange"] = "collective"
        if mode == "shm":
            gathered = ghost = None
            # the stage kernels store [sums | extrema] straight into the handle's pinned result buffer: no copy kernel
            rb, rb_dev = eng.result_buffer()
            hostn = rb[:row]
            vp_sums, vp_mm = C.c_void_p(rb_dev), C.c_void_p(rb_dev + 8 * ns)
            hp_sums = C.cast(hostn.ctypes.data, C.POINTER(C.c_double))
            hp_mm = C.cast(hostn.ctypes.data + 8 * ns, C.POINTER(C.c_double))

            def all_rows():
                stream.synchronize()
                return xchg.all_gather(hostn)
        else:
            gathered = torch.zeros(W * row, dtype=torch.float64, device="cuda")
            ghost = torch.zeros(W, row, dtype=torch.float64).pin_memory()
            g_np = ghost.numpy()

            def all_rows():
                dist.all_gather_into_tensor(gathered, buf, group=group)
                ghost.copy_(gathered.view(W, row), non_blocking=True)
                stream.synchronize()
                return g_np

        def gather_and_finish(E):
            g = all_rows()
            if E is None:
                E = g[:, ns:ns + 4].max(axis=0)
                if st["guess"] is None or not np.array_equal(E, st["guess"]):
                    return None, E                                              # the rows were weighted wrongly
            acc = g[0, :ns].copy()
            for r in range(1, W):                                               # rank order: same bits everywhere
                acc += g[r, :ns]
            own = g[rank, ns + 4:].copy()
            hostn[:ns] = acc
            hostn[ns:ns + 4] = E
            hostn[ns + 4:] = own
            return f3(h, hp_sums, hp_mm, op), E

        up = {"E": None}                                                        # extrema currently in `mmg`

        def stage2_with(E, emit=False):
            if up["E"] is None or not np.array_equal(E, up["E"]):              # upload the extrema only when they change
                e_np[:4] = E
                mmg.copy_(e_pin, non_blocking=True)
                up["E"] = np.array(E, np.float64)
            if emit:                                                            # rows + this shard's own extrema words
                return f2e(h, vp_mmg, vp_mm, vp_sums)
            return f2(h, vp_mmg, vp_sums)

        self.xchg = xchg                                                        # None unless the rows travel through shared memory
        if mode == "shm" and speculate:
            # the whole pass - stage 1, stage 2 with the guessed extrema, exchange, verification, finish - is one call
            # into the library (malio_measure_node), exactly what a C++ integration would use
            fnode, xh = lib.malio_measure_node, xchg.h
            stats2 = (C.c_int * 2)()

            def fn():
                rc = fnode(h, xh, sp, cv, op, stats2)
                st["hits"], st["misses"] = stats2[0] - base[0], stats2[1] - base[1]
                return rc
            base = list(eng.node_stats())
            fn._keep = (s, out, xchg, stats2)
            return fn, out
        if mode == "shm":
            # plain two-exchange sequence through shared memory (speculate=False): kept in Python, it is the reference
            # the speculating pass is tested against
            E_buf = np.zeros(4, np.float64)
            E_ptr = C.cast(E_buf.ctypes.data, C.POINTER(C.c_double))
            xr, xh, to = lib.malio_xchg_reduce, xchg.h, C.c_double(xchg.timeout)

            def fn():
                rc = f1(h, sp, cv, vp_mm)
                if rc < 0:
                    return rc
                E = all_rows()[:, ns:ns + 4].max(axis=0)                        # only the extrema words are valid yet
                rc = stage2_with(E)
                if rc < 0:
                    return rc
                stream.synchronize()
                r = xr(xh, hp_sums, ns, None, hp_sums, E_ptr, to)
                if r != 0:
                    raise capi.MalioError("malio_xchg_reduce rc=%d (a rank is missing?)" % r)
                hostn[ns:ns + 4] = E
                return f3(h, hp_sums, hp_mm, op)
            fn._keep = (s, out, mmg, e_pin, xchg, E_buf, rb)
            return fn, out

        def fn():
            spec = speculate and st["guess"] is not None
            # local extrema -> row[ns:ns+8]: by stage 1's fold launch, or (speculating) by stage 2 itself
            rc = f1(h, sp, cv, None if spec else vp_mm)
            if rc < 0:
                return rc
            if spec:
                rc = stage2_with(st["guess"], emit=True)
                if rc < 0:
                    return rc
                rc, E = gather_and_finish(None)
                if rc is not None:
                    st["hits"] += 1
                    return rc
                st["misses"] += 1
            else:
                dist.all_reduce(mm4, op=dist.ReduceOp.MAX, group=group)
                e_pin[:4].copy_(mm4, non_blocking=True)
                stream.synchronize()
                E = e_np[:4].copy()
            st["guess"] = E
            rc = stage2_with(E)
            if rc < 0:
                return rc
            rc, _ = gather_and_finish(E)
            return rc
        fn._keep = (s, out, buf, host, gathered, ghost, mmg, e_pin, xchg)
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
        x, t, converge, done, P_out = capi.ieskf_step(L, max_it, i, x, x_prop, P0, out["HtRinvH"], out["HtRinvh"], t,
                                                        limit=float(backend.params.get("limit", 0.0)))
        P = P_out  # the posterior when done, else the projected P_ the reference would be left with (esekfom.hpp:531-572)
        if done:
            break
    return dict(state=x, P=P, passes=passes, searches=searches, M=M, t=t)
