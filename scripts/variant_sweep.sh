#!/bin/bash
# Times the tuning builds under variants/ on c3 / c2 / c5 (scripts/prof_c3.py); usage: scripts/variant_sweep.sh [cfgs...]
cd "$(dirname "$0")/.."
for cfg in "${@:-c3 c5}"; do
  for lib in variants/libngicp_*.so; do
    echo "== $cfg $(basename $lib)"
    NGICP_LIB=$PWD/$lib timeout -k 10 120 python scripts/prof_c3.py 5 $cfg 2>&1 | grep "^align" | tail -2 || exit 1
  done
done
