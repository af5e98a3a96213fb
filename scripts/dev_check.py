"""Developer check on a GPU box: HIP engine vs CPU oracle, stage by stage, with timings.
Usage: python scripts/dev_check.py [n_small] [--big]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from direct_lidar_odometry_amd import clouds  # noqa: E402
from direct_lidar_odometry_amd.nano_gicp import NanoGICP  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def check_small(n):
    w = clouds.scan_to_scan(n)
    print(f"== small workload {w.name}: src {w.source.shape} tgt {w.target.shape}", flush=True)
    g = NanoGICP()
    o = orc.OracleGICP()
    for e in (g, o):
        e.setMaxCorrespondenceDistance(1.0)
        e.setInputSource(w.source)
        e.setInputTarget(w.target)
    # kNN
    ot = orc.OracleTree(w.target)
    for k in (1, 20):
        q = w.source[:: max(1, n // 2000)]
        gi, gd = g.target_knn(q, k)
        oi, od = ot.knn(q, k)
        print(f"knn k={k}: idx equal {np.array_equal(gi, oi)} ({(gi != oi).sum()} diffs), d2 equal {np.array_equal(gd, od)} max|dd| {np.abs(gd - od).max():.3e}", flush=True)
    # covariances
    g.calculateSourceCovariances(); g.calculateTargetCovariances()
    o.calculateSourceCovariances(); o.calculateTargetCovariances()
    for name, a, b in (("src", g.getSourceCovariances(), o.getSourceCovariances()), ("tgt", g.getTargetCovariances(), o.getTargetCovariances())):
        d = np.abs(a - b).reshape(len(a), -1).max(1)
        print(f"covs {name}: max abs diff {d.max():.3e}, #>1e-9: {(d > 1e-9).sum()} of {len(d)}", flush=True)
    print("stats", g.stats(), flush=True)
    # linearize
    T0 = np.eye(4)
    Hg, bg, eg = g.linearize(T0)
    Ho, bo, eo = o.linearize(T0)
    print(f"linearize: err {eg:.10e} vs {eo:.10e} rel {abs(eg - eo) / abs(eo):.2e}; H rel {np.abs(Hg - Ho).max() / np.abs(Ho).max():.2e}; b rel {np.abs(bg - bo).max() / np.abs(bo).max():.2e}", flush=True)
    cg, sg = g.correspondences()
    co, so = o.correspondences()
    print(f"correspondences: equal {np.array_equal(cg, co)} ({(cg != co).sum()} diffs), valid {np.mean(cg >= 0):.3f}", flush=True)
    T1 = clouds.make_pose((0.01, -0.02, 0.005), (0.1, 0.05, -0.1))
    print(f"compute_error: {g.compute_error(T1):.10e} vs {o.compute_error(T1):.10e}", flush=True)
    # align
    t = time.time(); g.align(); tg = time.time() - t
    t = time.time(); o.align(); to = time.time() - t
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    print("align hip   :", tg * 1e3, "ms; iters", g.nr_iterations_, "conv", g.converged_)
    print("align oracle:", to * 1e3, "ms; iters", o.nr_iterations, "conv", o.converged)
    print("pose diff hip vs oracle (m, rad):", clouds.pose_error(Tg, To), " vs GT:", clouds.pose_error(Tg, w.gt))
    print("trace hip\n", g.lm_trace())
    print("trace oracle\n", o.lm_trace())
    print("stats", g.stats(), flush=True)


def bench_big():
    for mk, name in ((lambda: clouds.scan_to_scan(100_000), "c2 s2s 100k/100k"), (lambda: clouds.scan_to_submap(100_000, 5), "c3 s2m 100k/500k")):
        w = mk()
        print(f"== {name}: src {w.source.shape} tgt {w.target.shape}", flush=True)
        g = NanoGICP()
        g.setMaxCorrespondenceDistance(w.max_corr_dist)
        g.setMaximumIterations(20); g.setTransformationEpsilon(1e-12); g.setRotationEpsilon(1e-12)
        t = time.time(); g.setInputTarget(w.target); print("  set target ms", (time.time() - t) * 1e3, g.stats()["index_build_ms"])
        t = time.time(); g.setInputSource(w.source); print("  set source ms", (time.time() - t) * 1e3, g.stats()["index_build_ms"])
        t = time.time(); g.calculateTargetCovariances(); print("  tgt covs ms", (time.time() - t) * 1e3, g.stats()["covariance_ms"])
        t = time.time(); g.calculateSourceCovariances(); print("  src covs ms", (time.time() - t) * 1e3, g.stats()["covariance_ms"])
        for rep in range(3):
            t = time.time(); g.align(w.guess); dt = time.time() - t
            s = g.stats()
            print(f"  align {dt * 1e3:.3f} ms loop {s['loop_ms']:.3f} ms passes {s['passes']} iters {s['outer_iterations']} -> {s['outer_iterations'] / s['loop_ms'] * 1e3:.0f} it/s  Cbar {s['mean_candidates']:.1f} valid {s['valid_fraction']:.3f} h {s['voxel_size']:.3f} grid {s['grid_dims']}", flush=True)
        print("  err vs GT", clouds.pose_error(g.getFinalTransformation(), w.gt), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10000
    check_small(n)
    if "--big" in sys.argv:
        bench_big()
