"""Developer aid: latency of the shared-memory all-gather (malio_xchg_reduce) between W processes of one host."""
import os, subprocess, sys, time, uuid
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    import ctypes as C
    from malio_amd import capi
    name, rank, W, row, n = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    if rank:
        while not os.path.exists("/dev/shm" + name):
            time.sleep(0.002)
        time.sleep(0.05)
    x = capi.NodeExchange(name, rank, W, row, create=(rank == 0), timeout_s=20.0)
    lib = capi.lib()
    r = np.arange(row, dtype=np.float64) + rank
    E = np.zeros(4)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ns = row - 8
    for _ in range(2000):
        lib.malio_xchg_reduce(x.h, p(r), ns, None, p(r), p(E), C.c_double(20.0))
    t = time.perf_counter()
    for _ in range(n):
        lib.malio_xchg_reduce(x.h, p(r), ns, None, p(r), p(E), C.c_double(20.0))
    dt = (time.perf_counter() - t) / n
    if rank == 0:
        x.unlink()
        print("W=%d row=%d doubles: %.2f us per gather+sum (incl. ~1 us of ctypes)" % (W, row, dt * 1e6), flush=True)
    x.close()
else:
    for W in (2, 4, 8):
        name = "/malio_time_" + uuid.uuid4().hex[:10]
        ps = [subprocess.Popen([sys.executable, __file__, "worker", name, str(r), str(W), "299", "50000"]) for r in range(W)]
        for q in ps:
            q.wait()
