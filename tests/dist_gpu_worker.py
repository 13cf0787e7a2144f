"""Worker of tests/test_gpu_parity.py::test_two_rank_sharded_pass_equals_single_engine (launched by torchrun, gloo
backend, both ranks on GPU 0): runs the pre-bound sharded pass several times and dumps what rank 0 got."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from malio_amd import capi, scenes  # noqa: E402
from malio_amd import dist as mdist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, W = dist.get_rank(), dist.get_world_size()
sc = scenes.make_scene(cfg=3)
scan_all = sc["scan"]
lo, hi = rank * sc["N"] // W, (rank + 1) * sc["N"] // W
eng = capi.Engine(sc["params"], device=0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.map_build(sc["map"])
eng.scan_set(scan_all[lo:hi], sc["tables"], sc["temporal_comp"])
be = mdist.HipBackend(eng)
res = {}
for label, spec in (("plain", False), ("spec", True)):
    fn, out = be.pass_fn(sc["state0"], True, speculate=spec)
    for _ in range(4):
        assert fn() >= 0
    Cc = eng.C
    res[label] = dict(M=int(out.M), w=float(out.w_loc), H=[float(x) for x in out.HtRinvH[:Cc * Cc]],
                      h=[float(x) for x in out.HtRinvh[:Cc]], stats=dict(be.spec_stats) if spec else None)
    if spec:
        res[label]["stats"]["guess"] = None
# a different state right after: the guess may or may not hold, the result must be exact either way
st2 = np.array(sc["state0"], np.float64).copy()
st2[:3] += (0.02, -0.01, 0.015)
fn2, out2 = be.pass_fn(st2, True, speculate=True)
for _ in range(3):
    assert fn2() >= 0
res["moved"] = dict(M=int(out2.M), H=[float(x) for x in out2.HtRinvH[:eng.C * eng.C]])
# the whole iterated update, sharded, in one library call per rank (shared-memory exchange; host-only hand-shake here)
from malio_amd.dist import exchange_mode  # noqa: E402
if exchange_mode() == "shm":
    name = "/malio_worker_upd_%s" % os.environ.get("MASTER_PORT", "0")
    row = eng.sums_len() + 8
    x = capi.NodeExchange(name, 0, W, row, create=True) if rank == 0 else None
    dist.barrier()
    if rank != 0:
        x = capi.NodeExchange(name, rank, W, row, create=False)
    dist.barrier()
    if rank == 0:
        x.unlink()
    eng.scan_set(scan_all[lo:hi], sc["tables"], sc["temporal_comp"])
    u = eng.update_iterated_node(x, sc["state0"], sc["P0"])
    res["update"] = dict(state=[float(v) for v in u["state"]], P00=float(u["P"][0, 0]), passes=u["passes"], M=u["M"])
    x.close()
allres = [None] * W
dist.all_gather_object(allres, res)
if rank == 0:
    json.dump(allres, open(sys.argv[1], "w"))
dist.barrier()
dist.destroy_process_group()
