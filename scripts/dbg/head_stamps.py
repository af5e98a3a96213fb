"""k_gicp_head's own stamps (NGICP_HEAD=1 NGICP_DEBUG_STAMPS=<file>): how long the head (state + subset rows + the optimiser's step) and the
tail (ticket, subset sum) take per wave.  usage: python scripts/dbg/head_stamps.py <file>"""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 24).astype(np.float64)
a = raw[(raw[:, 20] > 0) & (raw[:, 21] > 0)]
print("waves with a head:", len(a))
head = a[:, 21] - a[:, 20]
print("head (entry -> pose known) cycles p10/p50/p90/max:", np.percentile(head, [10, 50, 90, 100]).round(0))
m = a[:, 8] > 0
print("search + FP64 tail + block row (head end -> reduce done) p10/p50/p90/max:", np.percentile((a[m, 8] - a[m, 21]), [10, 50, 90, 100]).round(0))
m = (a[:, 22] > 0) & (a[:, 9] > 0)
print("barrier -> ticket drawn p10/p50/p90/max:", np.percentile((a[m, 22] - a[m, 9]), [10, 50, 90, 100]).round(0))
m = (a[:, 23] > 0) & (a[:, 22] > 0)
print("last blocks of a subset:", int(m.sum()), " ticket -> subset row stored p10/p50/p90/max:", np.percentile((a[m, 23] - a[m, 22]), [10, 50, 90, 100]).round(0) if m.any() else None)
life = np.where(a[:, 23] > 0, a[:, 23], np.where(a[:, 22] > 0, a[:, 22], a[:, 9])) - a[:, 20]
print("wave lifetime p10/p50/p90/max:", np.percentile(life, [10, 50, 90, 100]).round(0))
t0 = a[:, 20].min()
print("kernel span (first entry -> last stamp):", (np.maximum(a[:, 22], a[:, 23]).max() - t0).round(0), " entries p50/p90/max after the first:", np.percentile(a[:, 20] - t0, [50, 90, 100]).round(0))
