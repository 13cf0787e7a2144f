"""Developer aid (make PHASE=1 build): phase stamps of the workgroup that ended the last k_pass launch - the critical path
of the one-kernel pass.   MALIO_LIB=.../variants/phase.so python tools/pass_phase.py [cfg]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
e = capi.Engine(sc["params"]); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
names = ["enter", "point phase done", "extrema atomics", "row + LDS staging", "16 MFMA", "tile stores + vmcnt(0)", "group ticket + barrier",
         "group node (32 loads/thread)", "node -> host + vmcnt(0)", "global ticket", "extrema fold -> host"]
for conv in (True, False):
    acc = []
    for _ in range(30):
        e.measure(sc["state0"], conv)
        out = (C.c_longlong * 16)()
        assert capi.lib().malio_debug_pass_phase(out) == 0
        acc.append(np.array(out[:11], np.float64))
    t = np.median(np.diff(np.array(acc[5:]), axis=1), axis=0) / 100.0
    print("k_pass", "search" if conv else "reuse", "- last workgroup (us):")
    for n, d in zip(names[1:], t):
        print("   %-34s %6.2f" % (n, d))
    print("   %-34s %6.2f" % ("total in this workgroup", t.sum()))
print(e.fuse_stats())
