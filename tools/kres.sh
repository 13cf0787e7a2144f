#!/bin/bash
# Register / scratch / occupancy of every kernel of one .hip file (hipcc remarks; no GPU needed).
#   tools/kres.sh csrc/measure.hip [extra -D flags]
cd "$(dirname "$0")/../ma-lio_amd" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres_$$.o 2>&1 | sed 's/ \[-Rpass[^]]*\]//' |
  awk '/Function Name:/ {n=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /VGPRs Spill/ {sp=$NF} /LDS Size/ {printf "%s vgpr %s scratch %s spill %s occ %s lds %s\n", n, v, s, sp, o, $NF}' | c++filt | sed 's/(.*)//'
rm -f /tmp/kres_$$.o
