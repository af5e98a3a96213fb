"""Covariance kernel timing by regularisation mode / k (device time of the kernel from HIP events)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP
w = clouds.scan_to_submap(100_000, 5)
w5 = clouds.scan_to_submap(250_000, 2, shape="os1")
for name, cloud in (("vlp16 100k", w.source), ("vlp16 submap 500k", w.target), ("os1 250k", w5.source)):
    for k in (10, 20):
        for reg in (3, 0):
            e = NanoGICP(); e.setCorrespondenceRandomness(k); e.setRegularizationMethod(reg)
            e.setInputSource(cloud)
            ts = []
            for _ in range(4):
                e.calculateSourceCovariances(); ts.append(e.stats()["covariance_ms"])
            print(f"{name:18s} k={k:2d} reg={'PLANE' if reg == 3 else 'NONE '}: {min(ts[1:]):.3f} ms  ({min(ts[1:]) * 1e5 / len(cloud):.3f} ms per 100k)", flush=True)
            e.close()
