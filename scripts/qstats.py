"""Diagnostic: per-query search statistics + per-wave stamps of the LAST pass of an align (NGICP_DEBUG_QSTATS / NGICP_DEBUG_STAMPS).
usage: python scripts/qstats.py [c3|c2|c5]"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP, keyframe_covariances
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = clouds.scan_to_submap(100_000, 5) if cfg == "c3" else (clouds.scan_to_scan(100_000) if cfg == "c2" else clouds.scan_to_submap(250_000, 8, shape="os1"))
g = NanoGICP()
g.setMaxCorrespondenceDistance(w.max_corr_dist)
g.setMaximumIterations(20); g.setTransformationEpsilon(1e-12); g.setRotationEpsilon(1e-12)
if os.environ.get("QSTATS_COLD"):  # the FIRST pass of an alignment (no previous correspondences): one Gauss-Newton iteration
    g.setOptimizer(0); g.setMaximumIterations(1)
g.setInputTarget(w.target); g.setInputSource(w.source)
if cfg == "c2":
    g.calculateTargetCovariances()
else:
    g.setTargetCovariances(keyframe_covariances(w.target, w.keyframe_sizes, 20))
g.calculateSourceCovariances()
print("covariance_ms", g.stats()["covariance_ms"], "index_build_ms", g.stats()["index_build_ms"])
for _ in range(3):
    g.align(w.guess)
s = g.stats(); print(f"align {s['align_ms']:.3f} ms loop {s['loop_ms']:.3f} passes {s['passes']} Cbar {s['mean_candidates']:.1f} h {s['voxel_size']:.3f}")
os.makedirs("gpurun_out", exist_ok=True)
qf, sf = f"gpurun_out/qstats_{cfg}.bin", f"gpurun_out/stamps_{cfg}.bin"
os.environ["NGICP_DEBUG_QSTATS"] = qf
g.align(w.guess)
del os.environ["NGICP_DEBUG_QSTATS"]
q = np.fromfile(qf, dtype=np.int32).reshape(-1, 4)
c1, u, c2, fl = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
u1, uf = u & 0xffff, u >> 16
tot = c1 + c2
print("queries", len(q), "mean cand total", tot.mean(), "ring1", c1.mean(), "far+shell", c2.mean())
print("ring-1 walks per query: mean", u1.mean(), "hist", np.bincount(np.minimum(u1, 10)).tolist())
print("ring-1 cand per walk:", c1.sum() / max(1, u1.sum()), " windows/walk", c1.sum() / max(1, u1.sum()) / 12)
print("far walks per query: mean", uf.mean(), " queries with far walks", (uf > 0).mean(), " went_far", (fl & 1).mean(), " matched", ((fl & 4) > 0).mean())
print("total cand percentiles 10/50/90/99/100:", np.percentile(tot, [10, 50, 90, 99, 100]))
srt = np.sort(tot)[::-1]; cs = np.cumsum(srt) / tot.sum()
for f in (0.01, 0.05, 0.1, 0.25):
    print(f"  heaviest {f:.0%} of queries hold {cs[int(len(srt) * f)]:.1%} of the candidates")
um = (fl & 4) == 0
print("unmatched queries:", um.mean(), "their mean cand", tot[um].mean() if um.any() else 0, "share of all cand", tot[um].sum() / tot.sum())
os.environ["NGICP_DEBUG_SOLVE"] = "1"
g.align(w.guess)
del os.environ["NGICP_DEBUG_SOLVE"]
for k in (10, 20):
    e = NanoGICP(); e.setCorrespondenceRandomness(k)
    for name, cloud in (("source", w.source), ("target", w.target)):
        e.setInputSource(cloud)
        ts = []
        for _ in range(4):
            e.calculateSourceCovariances(); ts.append(e.stats()["covariance_ms"])
        print(f"covariances k={k} {name} n={len(cloud)}: {min(ts[1:]):.3f} ms")
    e.close()
os.environ["NGICP_DEBUG_STAMPS"] = sf
g.align(w.guess)
del os.environ["NGICP_DEBUG_STAMPS"]
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamps.py"), sf])
