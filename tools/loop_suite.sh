#!/bin/bash
# The GPU suite in a loop, the test FILES in a different (seeded) order every time - the round-5 probe-cache fault (a lane read
# another workgroup's LDS leftovers) showed up in 5 of 8 looped runs and in no single pass of the suite.
#   tools/loop_suite.sh <tag> [rounds=10]            the shipped library
#   tools/loop_suite.sh <tag> 1 poison               the -DMALIO_POISON build (ma-lio_amd/variants/poison.so: `make -C ma-lio_amd poison`)
#   tools/loop_suite.sh <tag> 1 cut                  MALIO_EARLY_MIN_QUERIES=0: every scan of every test ends its walks early
# Run via gpurun from the repo root; one line per round in gpurun_out/<tag>/<tag>_loop_<mode>.txt.
set -u
ulimit -c 0   # (an abort must not leave a core file of the process' whole address space on the box's disk)
TAG=$1; ROUNDS=${2:-10}; MODE=${3:-plain}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
case $MODE in
  poison) export MALIO_LIB=$ROOT/ma-lio_amd/variants/poison.so; [ -f $MALIO_LIB ] || { echo "no poison build"; exit 2; } ;;
  cut) export MALIO_EARLY_MIN_QUERIES=0 ;;
esac
LOG=$OUT/${TAG}_loop_${MODE}.txt
echo "# tools/loop_suite.sh $TAG $ROUNDS $MODE  (pytest -m gpu, test files shuffled with seed = round; MALIO_LIB=${MALIO_LIB:-shipped} MALIO_EARLY_MIN_QUERIES=${MALIO_EARLY_MIN_QUERIES:-default})" > $LOG
for r in $(seq 1 $ROUNDS); do
  FILES=$(python -c "import glob,random; f=sorted(glob.glob('tests/test_*.py')); random.Random($r).shuffle(f); print(' '.join(f))")
  timeout 1500 python -m pytest $FILES -m gpu -q --tb=short -rf -p no:cacheprovider > $OUT/loop_${MODE}_$r.log 2>&1
  rc=$?
  echo "round $r rc=$rc $(grep -E 'passed|failed|error' $OUT/loop_${MODE}_$r.log | tail -1) | order: $(echo $FILES | sed 's/tests\/test_//g; s/\.py//g')" >> $LOG
  [ $rc -ne 0 ] && grep -E "^(FAILED|ERROR)|Memory access fault|Aborted|core dumped" $OUT/loop_${MODE}_$r.log | head -20 >> $LOG
done
cat $LOG
