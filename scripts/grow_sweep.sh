#!/bin/bash
# align time vs the cap on listed rings (NGICP_STAGE_GROW); usage: scripts/grow_sweep.sh [cfgs...]
cd "$(dirname "$0")/.."
for cfg in "${@:-c3 c2}"; do
  for gr in 0 2 3 4 6; do
    echo "== $cfg stage_grow $gr"
    NGICP_STAGE_GROW=$gr timeout -k 10 120 python scripts/prof_c3.py 4 $cfg 2>&1 | grep "^align" | tail -1 || exit 1
  done
done
