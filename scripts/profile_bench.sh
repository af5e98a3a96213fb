#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command itself (the numbers bench.py's roofline block must agree with).
# Output: gpurun_out/prof_bench/...; scripts/summarize_bench_profile.py copies the summary into profiles/.
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
grep "^{" $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
echo bench profile done
