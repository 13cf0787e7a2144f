#!/bin/bash
# developer aid: kernel event times (tools/gpu_time.py) of every ma-lio_amd/variants/*.so, interleaved, 3 rounds
for rep in 1 2 3; do
for v in ma-lio_amd/variants/*.so; do
  echo -n "$(basename $v) CFG=${CFG:-2} "; MALIO_LIB=$PWD/$v python tools/gpu_time.py 2>/dev/null | grep KERNELS
done; done
