#!/bin/bash
# developer aid: where do k_pass' bytes and microseconds go?  Builds the attribution variants of csrc/measure.hip (see its
# ATTR_* switches: WRONG results, timing / counters only) next to the shipped build and, on the GPU box, prints per variant
# the kernel event times (tools/gpu_time.py, 3 interleaved rounds) and FETCH_SIZE / WRITE_SIZE of k_pass (own rocprofv3 runs).
#   CPU box:  tools/attr_variants.sh build
#   GPU box:  tools/attr_variants.sh run <outdir-under-gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = build ]; then
  shift
  exec bash $ROOT/tools/build_variants.sh csrc/measure.hip base:"" nosplit:"-DKS_SPLIT=0" nogather:"-DATTR_NO_GATHER" \
    nostate:"-DATTR_NO_STATE" walk5:"-DATTR_WALK5" notiles:"-DATTR_NO_TILES" "$@"
fi
OUT=$ROOT/gpurun_out/$2
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for v in ma-lio_amd/variants/*.so; do
    echo -n "$(basename $v .so) "; MALIO_LIB=$ROOT/$v python tools/gpu_time.py 2>/dev/null | grep KERNELS
  done
done | tee $OUT/attr_times.txt
cd /tmp && export TMPDIR=/tmp
for v in $ROOT/ma-lio_amd/variants/*.so; do
  n=$(basename $v .so)
  MALIO_LIB=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$n -o fetch -- python $ROOT/tools/gpu_time.py > $OUT/${n}_fetch_stdout.txt 2>&1
  MALIO_LIB=$v rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$n -o write -- python $ROOT/tools/gpu_time.py > $OUT/${n}_write_stdout.txt 2>&1
  echo "=== $n"
  python $ROOT/tools/pmc_summary.py $OUT/$n 2>&1 | grep -E "==|k_pass|k_search"
done | tee $OUT/attr_pmc.txt
# the raw counter CSVs are large: keep the summaries only
rm -rf $OUT/*/ 2>/dev/null
