#!/bin/bash
# developer aid: kernel event times (tools/gpu_time.py) of every ma-lio_amd/variants/*.so that is not a phase-clock build,
# interleaved, 3 rounds; CFG selects the BASELINE config (default 2)
for rep in 1 2 3; do
for v in ma-lio_amd/variants/*.so; do
  case $v in *phase*) continue;; esac
  echo -n "$(basename $v) CFG=${CFG:-2} "; MALIO_LIB=$PWD/$v python tools/gpu_time.py 2>/dev/null | grep KERNELS
done; done
