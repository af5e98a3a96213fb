"""Diagnostic: when and on which CU every block of the LAST pass of an align ran (NGICP_DEBUG_SPAN).
usage: python scripts/spans.py [c3|c2|c5] [iterations]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP, keyframe_covariances
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20  # the timeline is that of the LAST pass: fewer iterations look inside an alignment
w = clouds.scan_to_submap(100_000, 5) if cfg == "c3" else (clouds.scan_to_scan(100_000) if cfg == "c2" else clouds.scan_to_submap(250_000, 8, shape="os1"))
g = NanoGICP()
g.setMaxCorrespondenceDistance(w.max_corr_dist)
g.setMaximumIterations(iters); g.setTransformationEpsilon(1e-12); g.setRotationEpsilon(1e-12)
g.setInputTarget(w.target); g.setInputSource(w.source)
if cfg == "c2":
    g.calculateTargetCovariances()
else:
    g.setTargetCovariances(keyframe_covariances(w.target, w.keyframe_sizes, 20))
g.calculateSourceCovariances()
for _ in range(3):
    g.align(w.guess)
os.makedirs("gpurun_out", exist_ok=True)
path = f"gpurun_out/span_{cfg}.bin"
os.environ["NGICP_DEBUG_SPAN"] = path
g.align(w.guess)
del os.environ["NGICP_DEBUG_SPAN"]
s = g.stats(); print(f"align {s['align_ms']:.3f} ms loop {s['loop_ms']:.3f} passes {s['passes']}")
d = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
idx = np.nonzero(d[:, 1] > 0)[0]
d = d[idx]
st, en = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64)  # 10 ns ticks of the 100 MHz counter (the same on every CU)
t0 = st.min(); st -= t0; en -= t0
hw, xcc = (d[:, 2] & 0xffffffff).astype(np.int64), (d[:, 2] >> 32).astype(np.int64) & 0xf
cu = (xcc << 12) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
print("blocks", len(d), " CUs", len(np.unique(cu)), " XCCs", len(np.unique(xcc)), " span of the grid (first entry -> last exit) us:", en.max() / 100)
print("block life us p10/p50/p90/max:", (np.percentile(en - st, [10, 50, 90, 100]) / 100).tolist())
print("entry us p10/p50/p80/p90/max:", (np.percentile(st, [10, 50, 80, 90, 100]) / 100).tolist())
print("exit us p10/p50/p90/max:", (np.percentile(en, [10, 50, 90, 100]) / 100).tolist())
print("blocks resident at every us:", [int(np.sum((st <= t) & (en > t))) for t in np.arange(0, en.max(), 100)])
first = st < 300
print("first round:", int(first.sum()), "blocks, life p50/max", (np.percentile((en - st)[first], [50, 100]) / 100).tolist(),
      "; later rounds:", int((~first).sum()), "blocks" + (", entry p10/p50/p90 %s life p50/max %s" % ((np.percentile(st[~first], [10, 50, 90]) / 100).tolist(), (np.percentile((en - st)[~first], [50, 100]) / 100).tolist()) if (~first).any() else ""))
for i in np.argsort(en)[::-1][:8]:
    print("   late exit: blockIdx", int(idx[i]), "group", int(d[i, 3]), "entry", st[i] / 100, "exit", en[i] / 100, "life", (en[i] - st[i]) / 100)
per_cu = np.array([np.sum(cu == c) for c in np.unique(cu)])
print("blocks per CU min/p50/max:", int(per_cu.min()), int(np.median(per_cu)), int(per_cu.max()))
