"""What the PMC passes profile (tools/pmc_run.sh): 20 search passes of config CFG (default 2) with per-kernel event times.
SKIP unset / 0: every pass a FULL search (MALIO_OPT_SEARCH_SKIP off: every point walks its list) - the bench's step.
SKIP=1: the second-search-pass form: search-skip on, the state alternating between two iterates ~1.5 cm apart.
MALIO_OPT_PROBE_CACHE is OFF like in the bench's headline step (every point probes the directory) unless the environment sets
MALIO_PROBE_CACHE."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=int(os.environ.get("CFG", "2")))
skip = os.environ.get("SKIP", "0") == "1"
eng = capi.Engine(sc["params"]); eng.map_build(sc["map"]); eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
try:
    eng.set_option("search_skip", 1 if skip else 0)
    if "MALIO_PROBE_CACHE" not in os.environ:
        eng.set_option("probe_cache", 0)
except (AttributeError, RuntimeError):  # (a library of an earlier round, loaded through MALIO_LIB for an A/B: it has no options and never skips)
    pass
s2 = sc["state0"].copy(); s2[0:3] += [0.01, -0.008, 0.004]
states = [sc["state0"], s2] if skip else [sc["state0"]]
eng.measure(sc["state0"], True)
print("counters", eng.debug_counters())
eng.set_profiling(True)
acc = {}
for k in range(20):
    eng.measure(states[k % len(states)], True)
    for n, ms in eng.last_kernel_times(): acc.setdefault(n, []).append(ms * 1000)
print("KERNELS", {n: round(float(np.median(v)), 1) for n, v in acc.items()})
try:
    print("SKIP", eng.skip_stats())
except AttributeError:
    pass
