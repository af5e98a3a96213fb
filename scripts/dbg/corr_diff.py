import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP
from oracle import oracle as orc
w = clouds.scan_to_scan(10_000)
for gate in (None, 3.0):
    g, o = NanoGICP(), orc.OracleGICP()
    for e in (g, o):
        if gate: e.setMaxCorrespondenceDistance(gate)
        e.setInputSource(w.source); e.setInputTarget(w.target)
    g.calculateSourceCovariances(); g.calculateTargetCovariances()
    o.setSourceCovariances(g.getSourceCovariances()); o.setTargetCovariances(g.getTargetCovariances())
    for T in (np.eye(4), w.gt, clouds.make_pose((1.0, -2.0, 0.3), (3, -2, 25))):
        g.linearize(T); o.linearize(T)
        cg, sg = g.correspondences(); co, so = o.correspondences()
        d = np.flatnonzero(cg != co)
        print("gate", gate, "differ", len(d), "stats", g.stats()["mean_candidates"], g.stats()["voxel_size"], g.stats()["grid_dims"] if "grid_dims" in g.stats() else "")
        for i in d[:12]:
            q = (T[:3, :3] @ w.source[i].astype(np.float64) + T[:3, 3])
            print("  query", i, "q", q.round(3), "gpu idx", cg[i], "d2", sg[i], "oracle idx", co[i], "d2", so[i], "target gpu", w.target[cg[i]] if cg[i] >= 0 else None, "target oracle", w.target[co[i]] if co[i] >= 0 else None)
        print("  target bbox", w.target.min(0), w.target.max(0))
