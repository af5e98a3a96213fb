"""Group durations of the last pass (NGICP_DEBUG_COSTS dumps): the persistent kernel against one launch per pass.
usage (GPU box): NGICP_PERSIST=0 NGICP_DEBUG_COSTS=/tmp/c0.bin python scripts/prof_c3.py 3 c3; NGICP_PERSIST=1 NGICP_DEBUG_COSTS=/tmp/c1.bin python scripts/prof_c3.py 3 c3;
python scripts/dbg/cost_compare.py /tmp/c0.bin /tmp/c1.bin"""
import sys, numpy as np
out = []
for f in sys.argv[1:]:
    raw = np.fromfile(f, dtype=np.int32)
    # {costs[nb], order[nb], partials[nb][32] doubles}: nb from the size
    nb = len(raw) // (2 + 64)
    cost = raw[:nb].astype(np.float64) * 16  # cycles
    order = raw[nb:2 * nb]
    print(f, "groups", nb, "cost cycles p10/p50/p90/max:", np.percentile(cost, [10, 50, 90, 100]).round(0), "sum", cost.sum().round(0), "order is a permutation:", sorted(order.tolist()) == list(range(nb)),
          "first positions:", order[:6].tolist())
    out.append(cost)
if len(out) == 2 and len(out[0]) == len(out[1]):
    r = out[1] / np.maximum(out[0], 1)
    print("ratio second / first, per group p10/p50/p90:", np.percentile(r, [10, 50, 90]).round(2))
