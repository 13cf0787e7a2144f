"""developer aid: malio_predict_chain (three tracks on the device) against the same steps through malio_predict on the host"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
L = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sc = scenes.make_scene(seed=12, N=200, Nmap=3000, L=L)
eng = capi.Engine(sc["params"])
n = 17 + 6 * L
rng = np.random.default_rng(0)
x0 = sc["state0"]
P0 = np.ascontiguousarray(sc["P0"], np.float64)
A = rng.normal(size=(12, 12)); Q = np.ascontiguousarray(A @ A.T * 1e-4)
tt = np.arange(K) * 0.005
dt = np.full(K, 0.005); acc = np.ascontiguousarray(np.stack([np.sin(tt) * 2, np.cos(2 * tt), 9.8 + 0.3 * np.sin(3 * tt)], 1)); gy = np.ascontiguousarray(np.stack([0.3 * np.cos(tt), 0.2 * np.sin(2 * tt), np.full(K, 0.5)], 1))
# pre-built ctypes arguments for both paths
nt = 3
X = (capi.State * nt)(*[capi.state_from_flat(x0, L) for _ in range(nt)])
P = np.ascontiguousarray(np.stack([P0] * nt)); Kc = (C.c_int * nt)(K, K, K)
dts = np.ascontiguousarray(np.tile(dt, nt)); accs = np.ascontiguousarray(np.tile(acc, (nt, 1))); gys = np.ascontiguousarray(np.tile(gy, (nt, 1)))
out = (capi.State * (nt * K))()
f = capi.lib().malio_predict_chain
def dev():
    return f(eng.h, nt, X, capi._p(P, C.c_double), Kc, capi._p(dts, C.c_double), capi._p(accs, C.c_double), capi._p(gys, C.c_double), capi._p(Q, C.c_double), out)
g = capi.lib().malio_predict
xs = capi.state_from_flat(x0, L); Ph = P0.copy()
def host():
    for t in range(nt):
        for k in range(K):
            g(L, C.byref(xs), capi._p(Ph, C.c_double), C.c_double(0.005), capi._p(Q, C.c_double), capi._p(acc[k], C.c_double), capi._p(gy[k], C.c_double))
for name, fn in (("device chain (3 tracks x %d steps, one call)" % K, dev), ("host, %d malio_predict calls (incl. ctypes call overhead)" % (nt * K), host)):
    for _ in range(5): fn()
    ts = []
    for _ in range(30):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print("L=%d %-60s %.1f us" % (L, name, np.median(ts) * 1e6))
