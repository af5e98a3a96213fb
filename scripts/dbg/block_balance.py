"""How unevenly a block's four waves are loaded (walk kernel, NGICP_DEBUG_STAMPS): ring-1 units and lifetime per wave.
usage: NGICP_DEBUG_STAMPS=/tmp/st.bin python scripts/prof_c3.py 2 c3; python scripts/dbg/block_balance.py /tmp/st.bin"""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 24)
nb = len(raw) // 4
a = raw[: nb * 4].astype(np.float64).reshape(nb, 4, 24)
ok = (a[:, :, 0] > 0).all(axis=1)
a = a[ok]
units = a[:, :, 19]
life = a[:, :, 8] - a[:, :, 0]
search = a[:, :, 5] - a[:, :, 3]   # ring 1 + far rows
print("blocks", len(a))
for name, v in (("ring-1 units", units), ("wave lifetime (cycles)", life), ("search phase (cycles)", search)):
    mx, mean = v.max(axis=1), v.mean(axis=1)
    print(f"{name:26s}: per-wave p50/p90/max {np.percentile(v,50):9.0f} {np.percentile(v,90):9.0f} {v.max():9.0f} | block max p50/p90/max {np.percentile(mx,50):9.0f} {np.percentile(mx,90):9.0f} {mx.max():9.0f} | block mean p50/p90/max {np.percentile(mean,50):9.0f} {np.percentile(mean,90):9.0f} {mean.max():9.0f} | max/mean p50/p90 {np.percentile(mx/np.maximum(mean,1),50):.2f} {np.percentile(mx/np.maximum(mean,1),90):.2f}")
# if the four waves of a block shared their search work perfectly: the launch would be bounded by the largest block MEAN instead of the largest wave
order = np.argsort(-life.max(axis=1))[:8]
print("slowest blocks: wave lifetimes | units")
for b in order:
    print([int(x) for x in life[b]], [int(x) for x in units[b]])
