#!/bin/bash
# developer aid: tools/ab_pass.py on every ma-lio_amd/variants/*.so (but the poison / phase-clock builds), interleaved, REPS rounds
#   REPS=3 CFG=2 tools/ab_variants.sh > gpurun_out/<tag>/ab.txt
for rep in $(seq 1 ${REPS:-3}); do
for v in ma-lio_amd/variants/*.so; do
  case $v in *poison*|*phase*|*nopcmark*) continue;; esac
  MALIO_LIB=$PWD/$v python tools/ab_pass.py 2>/dev/null | tail -1
done; done
