"""Timing of a small (DLO-after-voxel-filter sized) scan-to-submap alignment: 20k -> 100k points, DLO's settings and fixed 20 iterations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP
w = clouds.scan_to_submap(20_000, 5)
for name, it, eps in (("fixed 20", 20, 1e-12), ("DLO 32 / 0.01", 32, 0.01)):
    g = NanoGICP(); g.setMaxCorrespondenceDistance(w.max_corr_dist); g.setMaximumIterations(it); g.setTransformationEpsilon(eps); g.setRotationEpsilon(2e-3 if eps > 1e-6 else 1e-12)
    g.setInputTarget(w.target); g.setInputSource(w.source); g.calculateTargetCovariances(); g.calculateSourceCovariances()
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); g.align(w.guess); ts.append((time.perf_counter() - t0) * 1e3)
    s = g.stats()
    print(f"{name}: n_src {len(w.source)} n_tgt {len(w.target)} align wall ms {min(ts[1:]):.3f} engine {s['align_ms']:.3f} loop {s['loop_ms']:.3f} passes {s['passes']} -> {1e3 * s['loop_ms'] / s['passes']:.1f} us per pass+solve; index {s['index_build_ms']:.3f} cov {s['covariance_ms']:.3f}")
    g.close()
