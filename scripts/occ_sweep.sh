#!/bin/bash
# align time vs target own-cell occupancy of the automatic voxel sizing; usage: scripts/occ_sweep.sh [cfgs...]
cd "$(dirname "$0")/.."
for cfg in "${@:-c3 c5}"; do
  for occ in 8 12 16 20 24 32 48; do
    echo "== $cfg occ $occ"
    NGICP_TARGET_OCC=$occ timeout -k 10 120 python scripts/prof_c3.py 5 $cfg 2>&1 | grep "^align" | tail -1 || exit 1
  done
done
