#!/bin/bash
# rocprofv3 profiles of one workload (scripts/prof_c3.py <reps> <cfg>): kernel trace + stats, then PMC counters in their own
# passes.  Every PMC pass also carries --kernel-trace (allowed next to --pmc; no sys / runtime / hip / hsa / memory-copy tracing is
# mixed in, as the pool requires): scripts/pmc_summary.py joins counter rows to launch durations by dispatch id inside each pass.  usage: scripts/pmc_profile.sh <cfg: c3|c2|c5> [reps]
# Output: gpurun_out/pmc_<cfg>/<pass>/...; summarise with scripts/pmc_summary.py <cfg>.
set -o pipefail
CFG=${1:-c3}; REPS=${2:-3}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$CFG
[ -z "$PMC_ONLY" ] && rm -rf $OUT; mkdir -p $OUT
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/scripts/prof_c3.py $REPS $CFG > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  echo "pass $name ok"
}
if [ -z "$PMC_ONLY" ]; then run trace --stats || exit 1; fi
PASSES=(
  "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
  "sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
  "ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
  "ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  "tcp1 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
  "tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
  "tcp3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
  "td1 TD_TD_BUSY_sum TD_TC_STALL_sum"
  "tcp4 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  "l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  "fetch FETCH_SIZE"
  "write WRITE_SIZE"
  "grbm GRBM_GUI_ACTIVE GRBM_COUNT"
)
for p in "${PASSES[@]}"; do
  set -- $p
  name=$1; shift
  if [ -n "$PMC_ONLY" ] && [[ " $PMC_ONLY " != *" $name "* ]]; then continue; fi
  run $name --pmc "$@"   # a pass the hardware cannot schedule is reported and skipped
done
echo profiles done
