#!/bin/bash
# developer aid: HBM fetch / write counters of the search-pass kernels for every ma-lio_amd/variants/*.so
#   tools/pmc_variants.sh <outdir-under-gpurun_out>      (counters in their own runs, kernel-trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $ROOT/ma-lio_amd/variants/*.so; do
  n=$(basename $v .so)
  MALIO_LIB=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$n -o fetch -- python $ROOT/tools/gpu_time.py > $OUT/${n}_fetch_stdout.txt 2>&1
  MALIO_LIB=$v rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$n -o write -- python $ROOT/tools/gpu_time.py > $OUT/${n}_write_stdout.txt 2>&1
  MALIO_LIB=$v rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/$n -o tcc -- python $ROOT/tools/gpu_time.py > $OUT/${n}_tcc_stdout.txt 2>&1
  echo "=== $n"
  python $ROOT/tools/pmc_summary.py $OUT/$n 2>&1 | grep -E "==|k_pass"
done
