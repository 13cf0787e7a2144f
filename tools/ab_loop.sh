#!/bin/bash
for v in ma-lio_amd/variants/*.so; do
  for cfgi in 2 5; do
    MALIO_LIB=$PWD/$v python bench.py --config $cfgi --no-cpu-baseline --steps 300 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v cfg$cfgi', round(d['ms_per_step']*1e3,2), 'us', {k: round(v*1e3,1) for k,v in d['roofline']['kernel_event_ms'].items()}, 'loop', {k: round(v,3) for k,v in d['secondary']['scan_loop'].items() if k.endswith('_ms')})"
  done
done
