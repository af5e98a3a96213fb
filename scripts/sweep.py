"""Sweep engine knobs (lanes per query, voxel size) on one workload; prints loop time per setting."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
lanes_list = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["4", "8", "16"])]
vox_list = [float(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "0.15", "0.25", "0.4"])]
w = {"c3": lambda: clouds.scan_to_submap(100_000, 5), "c2": lambda: clouds.scan_to_scan(100_000),
     "c5": lambda: clouds.scan_to_submap(250_000, 8, shape="os1")}[cfg]()
ref = None
covs = None
for vox in vox_list:
    for lanes in lanes_list:
        g = NanoGICP()
        g.setTuning(vox, lanes)
        g.setMaxCorrespondenceDistance(w.max_corr_dist)
        g.setMaximumIterations(20); g.setTransformationEpsilon(1e-12); g.setRotationEpsilon(1e-12)
        g.setInputTarget(w.target); g.setInputSource(w.source)
        if covs is None:
            g.calculateTargetCovariances(); g.calculateSourceCovariances()
            covs = (g.getSourceCovariances(), g.getTargetCovariances())
        else:
            g.setSourceCovariances(covs[0]); g.setTargetCovariances(covs[1])
        best = 1e9
        for r in range(4):
            g.align(w.guess); s = g.stats(); best = min(best, s["loop_ms"])
        T = g.getFinalTransformation()
        if ref is None: ref = T
        print(f"{cfg} vox {vox:5.2f} (h={s['voxel_size']:.3f} grid {s['grid_dims']}) lanes {lanes:2d}: loop {best:8.3f} ms  passes {s['passes']} iters {s['outer_iterations']} "
              f"-> {s['outer_iterations'] / best * 1e3:8.0f} it/s  Cbar {s['mean_candidates']:.1f} valid {s['valid_fraction']:.3f} align {s['align_ms']:.3f} ms  dT {np.abs(T - ref).max():.2e}", flush=True)
        g.close()
