#!/bin/bash
# developer aid: bench the kernel variants under ma-lio_amd/variants/*.so back to back on one box
for rep in 1 2 3; do
for v in ma-lio_amd/variants/*.so; do
    MALIO_LIB=$PWD/$v python bench.py --no-cpu-baseline --steps 400 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1e3,2), 'us', {k: round(v*1e3,1) for k,v in d['roofline']['kernel_event_ms'].items()}, 'update', round(d['eskf']['update_ms'],3))"
done; done
