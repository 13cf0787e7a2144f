"""malio_xchg_*: the shared-memory all-gather the ranks of one node use for the per-pass results (host only)."""
import os
import subprocess
import sys
import uuid

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "xchg_worker.py")


def run(world, row, epochs, die_at=-1):
    name = "/malio_test_" + uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, WORKER, name, str(r), str(world), str(row), str(epochs), str(die_at)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=180)
        outs.append((p.returncode, o))
    if os.path.exists("/dev/shm" + name):  # only when the creator was killed
        os.unlink("/dev/shm" + name)
    return outs


@pytest.mark.parametrize("world,row", [(2, 299), (4, 105), (8, 299)])
def test_all_gather_in_rank_order_and_in_step(capi, world, row):
    outs = run(world, row, 3000)
    accs = set()
    for rc, o in outs:
        assert rc == 0 and "OK rank" in o, o
        accs.add(o.strip().split("acc ")[1])
        red = [ln for ln in o.splitlines() if ln.startswith("REDUCE")][0].split()
        assert red[3:] == ["1", "0", "0", "True", "True", "True"], o   # miss reported, hit summed, in-place summed
    assert len(accs) == 1  # every rank formed the same sums, bit for bit


def test_missing_rank_is_an_error_not_a_hang(capi):
    outs = run(3, 16, 1000, die_at=400)
    survivors = [o for rc, o in outs[:-1]]
    assert all("TIMEOUT" in o and "epoch 400" in o for o in survivors), outs


def test_bad_arguments(capi):
    import ctypes as C
    lib = capi.lib()
    h = C.c_void_p()
    assert lib.malio_xchg_create(b"no_slash", 0, 2, 8, 1, C.byref(h)) == capi.ERR_BAD_ARG
    assert lib.malio_xchg_create(b"/malio_x", 2, 2, 8, 1, C.byref(h)) == capi.ERR_BAD_ARG
    assert lib.malio_xchg_create(b"/malio_does_not_exist_%d" % os.getpid(), 1, 2, 8, 0, C.byref(h)) != capi.OK
    assert lib.malio_xchg_destroy(None) == capi.OK
