#!/bin/bash
# The first box with two or more GPUs: run THIS before the scaling bench, so that the first SCALE run is not also the first time
# RCCL sees more than one rank of this library.
#   tools/first_multi_gpu.sh [n_gpus=2]       (from the repo root; writes gpurun_out/first_multi_gpu/)
# 1. which librccl / libamdhip64 does a process that imports torch and loads libmalio_hip.so end up with (one copy of each is
#    the rule: ma-lio_amd/capi.py preloads torch's; a C++ host without torch resolves through the library's RUNPATH, /opt/rocm/lib)
# 2. the two pytest cases that are skipped below two devices (malio_node_create(n_gpus = 2, XCHG_RCCL), both partitionings)
# 3. bench.py --gpus N over RCCL, short, with the watchdog's limit lowered - its JSON line says `exchange` / `rccl_ranks`, carries
#    the measured curve's point next to `predicted` (the one-GPU proxy) - and once more with MALIO_EARLY_MIN_QUERIES=0
set -u
N=${1:-2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/first_multi_gpu
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python - > $OUT/libraries.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi
capi.lib()
print("torch", torch.__version__, "| devices visible to the library:", capi.lib().malio_device_count(), "| build", capi.lib().malio_build_id().decode())
seen = set()
for line in open("/proc/self/maps"):
    p = line.split()[-1]
    if any(k in p for k in ("librccl", "libamdhip64", "libmalio_hip", "libhsa-runtime")) and p not in seen:
        seen.add(p); print("mapped:", p)
n = {k: sum(1 for p in seen if k in p) for k in ("librccl", "libamdhip64")}
print("copies:", n, "-> OK" if all(v == 1 for v in n.values()) else "-> MORE THAN ONE COPY of a runtime in this process: fix the search path before trusting anything below")
PY
cat $OUT/libraries.txt
timeout 900 python -m pytest tests/test_partition.py -q -m gpu -k "two_gpus_over_rccl" 2>&1 | tail -5 | tee $OUT/pytest_two_gpus.txt
MALIO_RCCL_LEG_TIMEOUT_S=120 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29711 \
  bench.py --gpus $N --steps 50 --warmup 5 > $OUT/bench_gpus$N.json 2> $OUT/bench_gpus$N.stderr
# the same with the shards allowed to end their list walks early whatever they serve (MALIO_OPT_EARLY_MIN_QUERIES = 0: a shard of
# eight serves 25 k of the scan's 200 k points, below the default threshold of 32 768 - even on the one-GPU proxy,
# profiles/round6/r06_virtual_shards_cut.json; this is where it meets real GPUs)
MALIO_EARLY_MIN_QUERIES=0 MALIO_RCCL_LEG_TIMEOUT_S=120 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29712 \
  bench.py --gpus $N --steps 50 --warmup 5 > $OUT/bench_gpus${N}_cut.json 2> $OUT/bench_gpus${N}_cut.stderr
python - <<PY
import json
try:
    jc = json.loads([l for l in open("$OUT/bench_gpus${N}_cut.json") if l.startswith("{")][-1])
    print("bench --gpus $N, MALIO_EARLY_MIN_QUERIES=0: exchange", jc.get("exchange"), "rccl_ranks", jc.get("rccl_ranks"), "ms_per_step %.4f" % jc["ms_per_step"])
except Exception as e:
    print("no bench line with MALIO_EARLY_MIN_QUERIES=0:", e)
try:
    js = json.loads([l for l in open("$OUT/bench_gpus$N.json") if l.startswith("{")][-1])
    print("bench --gpus $N: exchange", js.get("exchange"), "rccl_ranks", js.get("rccl_ranks"), "ms_per_step %.4f" % js["ms_per_step"], js.get("rccl_note", ""))
    print("   variants:", {k: round(v["ms_per_step"], 4) for k, v in js["variants"].items()})
    print("   replicas:", js.get("replicas"))
    print("   predicted:", js.get("predicted"))
except Exception as e:
    print("no bench line:", e)
PY
