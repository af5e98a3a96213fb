"""Histogram of dependent window steps per (query, row) unit of the LAST pass (NGICP_DEBUG_QSTATS): usage: python scripts/dbg/step_hist.py [c3|c5]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["NGICP_DEBUG_QSTATS"] = "/tmp/qs.bin"
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = clouds.scan_to_submap(100_000, 5) if cfg == "c3" else clouds.scan_to_submap(250_000, 8, shape="os1")
g = NanoGICP(); g.setMaxCorrespondenceDistance(w.max_corr_dist)
g.setMaximumIterations(12); g.setTransformationEpsilon(1e-12); g.setRotationEpsilon(1e-12)
g.setInputTarget(w.target); g.setInputSource(w.source); g.calculateTargetCovariances(); g.calculateSourceCovariances()
g.align(w.guess)
n = len(w.source)
raw = np.fromfile("/tmp/qs.bin", dtype=np.int32)
h = raw[n * 4: n * 4 + 128]
for name, hh in (("ring-1 units", h[:64]), ("listed-row units", h[64:128])):
    tot = hh.sum(); c = np.cumsum(hh) / max(1, tot)
    print(name, "units", int(tot), "per wave", round(tot / (n / 28.9), 1), "| steps histogram", hh[:24].tolist(), "| share with > 2/3/4/6/8 steps:", [round(float(1 - c[k]), 4) for k in (2, 3, 4, 6, 8)],
          "| steps beyond a budget of 3/4/6 (sum):", [int(sum(max(0, s - b) * int(hh[s]) for s in range(64))) for b in (3, 4, 6)])
