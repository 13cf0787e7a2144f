#!/bin/bash
# PMC passes for the bench command (counters in their own runs, kernel-trace only: see the gpurun rules).
# usage: tools/pmc_run.sh <outdir-under-gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/gpu_time.py"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT -o sq -- $CMD > $OUT/sq_stdout.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT -o tcc -- $CMD > $OUT/tcc_stdout.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $CMD > $OUT/fetch_stdout.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT -o write -- $CMD > $OUT/write_stdout.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT -o tcp -- $CMD > $OUT/tcp_stdout.txt 2>&1
ls $OUT
