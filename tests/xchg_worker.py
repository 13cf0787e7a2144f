"""Worker of tests/test_node_exchange.py: one rank of a shared-memory exchange (host only, no GPU)."""
import os
import sys
import time

import numpy as np

os.environ["MALIO_TORCH_FIRST"] = "0"  # host-only process: never imports torch (see capi._share_hip_runtime_with_torch)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from malio_amd import capi  # noqa: E402


def main():
    name, rank, world, row, epochs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    die_at = int(sys.argv[6]) if len(sys.argv) > 6 else -1
    if rank != 0:  # the creator must be first: wait for the segment to appear
        t0 = time.time()
        while not os.path.exists("/dev/shm" + name):
            if time.time() - t0 > 30:
                sys.exit(3)
            time.sleep(0.005)
        time.sleep(0.05)
    x = capi.NodeExchange(name, rank, world, row, create=(rank == 0), timeout_s=3.0)
    first = True
    mine = np.zeros(row, np.float64)
    acc = 0.0
    for e in range(1, epochs + 1):
        if e == die_at and rank == world - 1:
            os._exit(0)  # a rank disappears: the others must get an error, not hang
        mine[:] = rank * 1000.0 + e + np.arange(row) * 1e-3
        try:
            g = x.all_gather(mine)
            if first and rank == 0:   # everybody took part in an exchange, so everybody has the segment mapped
                x.unlink()
                assert not os.path.exists("/dev/shm" + name)
            first = False
        except capi.MalioError:
            print("TIMEOUT rank %d epoch %d" % (rank, e), flush=True)
            x.close()
            sys.exit(0)
        want = (np.arange(world)[:, None] * 1000.0 + e) + np.arange(row)[None, :] * 1e-3
        if not np.array_equal(g, want):
            print("MISMATCH rank %d epoch %d" % (rank, e), flush=True)
            sys.exit(2)
        s = g[0].copy()
        for r in range(1, world):  # rank order: the same bits on every rank
            s += g[r]
        acc += float(s.sum())
    # malio_xchg_reduce: row = [ns sums | 4 extrema words | ...]; a right guess gives the rank-ordered sums, a wrong one
    # is reported with the true extrema and leaves the sums alone
    import ctypes as C
    lib = capi.lib()
    ns = row - 8
    if ns > 0:
        r_in = np.zeros(row, np.float64)
        r_in[:ns] = (rank + 1) * (1.0 + np.arange(ns) * 0.25)
        r_in[ns:ns + 4] = [rank, -rank, 10.0 - rank, 0.5]
        E_true = np.array([world - 1, 0.0, 10.0, 0.5])
        sums = np.full(ns, -1.0)
        E = np.zeros(4)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        wrong = E_true + [0, 0, 1e-9, 0]
        rc1 = lib.malio_xchg_reduce(x.h, p(r_in), ns, p(wrong), p(sums), p(E), C.c_double(3.0))
        untouched = bool(np.all(sums == -1.0)) and np.array_equal(E, E_true)
        rc2 = lib.malio_xchg_reduce(x.h, p(r_in), ns, p(E_true), p(sums), p(E), C.c_double(3.0))
        want = sum((r + 1) for r in range(world)) * (1.0 + np.arange(ns) * 0.25)
        rc3 = lib.malio_xchg_reduce(x.h, p(r_in), ns, None, p(r_in), p(E), C.c_double(3.0))   # in place, no check
        print("REDUCE rank %d %d %d %d %s %s %s" % (rank, rc1, rc2, rc3, untouched, np.array_equal(sums, want),
                                                    np.array_equal(r_in[:ns], want)), flush=True)
    print("OK rank %d acc %r" % (rank, acc), flush=True)
    x.close()


if __name__ == "__main__":
    main()
