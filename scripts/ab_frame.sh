#!/bin/bash
# A/B of the bench's frame-level figures under an environment setting; usage: scripts/ab_frame.sh VAR=VALUE [reps]
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${2:-3}); do for setting in "" "$1"; do
  echo "== ${setting:-default}"
  env $setting timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['dlo_frame']
print(round(d['value']), 'ms/scan', round(d['ms_per_scan'],4), 'frame', round(f['frame_ms'],4), 's2s', round(f['scan_to_scan_align_ms'],4), 's2m', round(f['scan_to_submap_align_ms'],4), 'c5', round(d['c5']['iterations_per_s']))"
done; done
