#!/bin/bash
# Profiles of the bench workload on the GPU box (run through gpurun).  Outputs under gpurun_out/prof_r01/.
# PMC counters are collected in their own passes (no tracing domains mixed in), as the pool requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r01
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_l2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_sq.log 2>&1 || exit 1
echo profiles done
