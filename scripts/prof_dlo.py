"""DLO-settings aligns (eps 0.01, <= 32 iterations): where does a short align spend its time?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP
w = clouds.scan_to_submap(100_000, 5)
g = NanoGICP()
g.setCorrespondenceRandomness(20); g.setMaxCorrespondenceDistance(0.5); g.setMaximumIterations(32); g.setTransformationEpsilon(0.01)
g.setInputTarget(w.target); g.setInputSource(w.source)
g.calculateTargetCovariances(); g.calculateSourceCovariances()
for r in range(5):
    t0 = time.perf_counter(); g.align(w.guess); t1 = time.perf_counter(); s = g.stats()
    print(f"wall {1e3*(t1-t0):.3f} ms align {s['align_ms']:.3f} loop {s['loop_ms']:.3f} passes {s['passes']} iters {s['outer_iterations']} -> {1e3*s['loop_ms']/max(1,s['passes']):.1f} us/pass  Cbar {s['mean_candidates']:.1f}", flush=True)
