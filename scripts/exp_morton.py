"""Experiment: does query ordering (spatial compactness of a wave's batch) matter?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP

def morton_order(p, cell):
    q = np.floor((p - p.min(0)) / cell).astype(np.uint64)
    def spread(v):
        v = v & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")

w = clouds.scan_to_submap(100_000, 5)
base = NanoGICP(); base.setInputTarget(w.target); base.setInputSource(w.source)
base.calculateTargetCovariances(); base.calculateSourceCovariances()
cs, ct = base.getSourceCovariances(), base.getTargetCovariances()
for name, order, srcvox in (("linear (engine order)", None, 0.0), ("morton 0.1 + src vox 2m", morton_order(w.source, 0.1), 2.0), ("morton 0.1 + src vox 4m", morton_order(w.source, 0.1), 4.0), ("random + src vox 4m", np.random.default_rng(0).permutation(len(w.source)), 4.0)):
    src = w.source if order is None else np.ascontiguousarray(w.source[order])
    cov = cs if order is None else np.ascontiguousarray(cs[order])
    for lanes in (2, 4, 8):
        g = NanoGICP(); g.setTuning(0.0, lanes)
        g.setMaxCorrespondenceDistance(w.max_corr_dist); g.setMaximumIterations(20); g.setTransformationEpsilon(1e-12); g.setRotationEpsilon(1e-12)
        g.setInputTarget(w.target); g.setTargetCovariances(ct)
        g.setTuning(srcvox, lanes); g.setInputSource(src); g.setSourceCovariances(cov)
        best = 1e9
        for r in range(4):
            g.align(w.guess); s = g.stats(); best = min(best, s["loop_ms"])
        print(f"{name:28s} lanes {lanes}: loop {best:7.3f} ms passes {s['passes']} Cbar {s['mean_candidates']:.1f} T00 {g.getFinalTransformation()[0,3]:.6f}", flush=True)
        g.close()
