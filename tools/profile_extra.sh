#!/bin/bash
# second half of a profile round: GPU test summary, every BASELINE config, one turn of the mapping loop as a timeline,
# the update-mode comparison. usage (via gpurun, repo root): tools/profile_extra.sh r02c
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $OUT/${TAG}_gputest.txt
rm -f $OUT/${TAG}_configs.jsonl
for c in 1 3 4 5 2; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_configs.jsonl; done
python tools/probe_jitter.py 2 200 2>/dev/null | grep -E "^(host|gated)" > $OUT/${TAG}_update_modes.txt
python tools/probe_devloop.py 2 5 2>/dev/null | grep -vE "Rebuild|Multi" >> $OUT/${TAG}_update_modes.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/loop -o loop -- python $ROOT/tools/time_mapinc.py > $OUT/loop_stdout.txt 2>&1
( echo "# one turn of the mapping loop (scan_set -> update_iterated -> map_incremental), config 2"
  echo "# rocprofv3 --kernel-trace --memory-copy-trace -- python tools/time_mapinc.py ; python tools/loop_timeline.py <kernel csv> <copy csv> 7"
  python $ROOT/tools/loop_timeline.py $OUT/loop/loop_kernel_trace.csv $OUT/loop/loop_memory_copy_trace.csv 7 ) > $OUT/${TAG}_loop_timeline.txt 2>&1
# the pipelined loop (next scan staged ahead, packed records): per-call times without the tracer, then its timeline under it
( echo "# python tools/time_pipeline.py (8 turns back to back per line; us per turn and per call on the calling thread)"
  python $ROOT/tools/time_pipeline.py 2>/dev/null | grep -E "^(turn|fuse)"
  echo "# POINTS=1 (48-byte points)"
  POINTS=1 python $ROOT/tools/time_pipeline.py 2>/dev/null | grep -E "^turn"
  echo "# MALIO_MAINT_STREAM=0"
  MALIO_MAINT_STREAM=0 python $ROOT/tools/time_pipeline.py 2>/dev/null | grep -E "^turn"
  echo "# MALIO_MAPINC_SMALL=0"
  MALIO_MAPINC_SMALL=0 python $ROOT/tools/time_pipeline.py 2>/dev/null | grep -E "^turn" ) > $OUT/${TAG}_pipeline_turn.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/pipe -o pipe -- python $ROOT/tools/time_pipeline.py > $OUT/pipe_stdout.txt 2>&1
python $ROOT/tools/pipeline_timeline.py $(find $OUT/pipe -name "pipe_kernel_trace.csv") $(find $OUT/pipe -name "pipe_memory_copy_trace.csv") > $OUT/${TAG}_pipeline_timeline.txt 2>&1
cat $OUT/${TAG}_gputest.txt; tail -3 $OUT/${TAG}_loop_timeline.txt; cat $OUT/${TAG}_pipeline_turn.txt
