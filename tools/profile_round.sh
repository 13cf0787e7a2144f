#!/bin/bash
# One GPU call that produces everything profiles/ keeps for a tag: tools/profile_round.sh r01d
# (run via gpurun from the repo root; outputs under gpurun_out/<tag>/)
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/${TAG}_bench_stdout.txt 2> $OUT/bench_stderr.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench -- python $ROOT/bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof_stdout.txt 2> $OUT/rocprof_stderr.txt
bash $ROOT/tools/pmc_run.sh $TAG/pmc > $OUT/pmc_run_stdout.txt 2>&1
python $ROOT/tools/pmc_summary.py $OUT/pmc > $OUT/${TAG}_pmc_summary.txt 2>&1
python $ROOT/tools/pmc_traffic.py $OUT $TAG k_pass round6 > $OUT/pmc_traffic_stdout.txt 2>&1
# the three-kernel pass on the same box, for comparison (MALIO_FUSE=0)
MALIO_FUSE=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_nofuse -- python $ROOT/tools/run_passes.py > /dev/null 2>&1
python $ROOT/tools/pass_timeline.py $OUT/${TAG}_nofuse_kernel_trace.csv > $OUT/${TAG}_pass_timeline_three_kernel.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o ${TAG}_fuse -- python $ROOT/tools/run_passes.py > /dev/null 2>&1
python $ROOT/tools/pass_timeline.py $OUT/${TAG}_fuse_kernel_trace.csv > $OUT/${TAG}_pass_timeline.txt 2>&1
ls $OUT
tail -c 600 $OUT/${TAG}_bench_stdout.txt
