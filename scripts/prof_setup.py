"""Profile target for the setup path: index builds (100k scan, 500k and 2M submaps through the device keyframe store) and covariances.
usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/prof_setup.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_lidar_odometry_amd import clouds
from direct_lidar_odometry_amd.nano_gicp import NanoGICP
for cfg in ("c3", "c5"):
    w = clouds.scan_to_submap(100_000, 5) if cfg == "c3" else clouds.scan_to_submap(250_000, 8, shape="os1")
    s2s, g = NanoGICP(), NanoGICP()
    s2s.setCorrespondenceRandomness(20)
    lo = 0
    for n in w.keyframe_sizes:
        s2s.setInputSource(np.ascontiguousarray(w.target[lo:lo + n])); s2s.calculateSourceCovariances(); g.addKeyframe(s2s); lo += n
    ids = list(range(len(w.keyframe_sizes)))
    for rep in range(4):
        g.setSubmapKeyframes(ids[:-1]); g.stats()
        t0 = time.perf_counter(); g.setSubmapKeyframes(ids); s = g.stats(); t1 = time.perf_counter()
        print(f"{cfg} submap {len(w.target)} points: host wall {1e3 * (t1 - t0):.3f} ms, submap_ms {s['submap_ms']:.3f}, index_build_ms {s['index_build_ms']:.3f}", flush=True)
    e = NanoGICP()
    for rep in range(3):
        tgt = np.ascontiguousarray(w.target + np.float32(1e-4 * rep))
        t0 = time.perf_counter(); e.setInputTarget(tgt); s = e.stats(); t1 = time.perf_counter()
        print(f"{cfg} setInputTarget {len(tgt)} points: host wall {1e3 * (t1 - t0):.3f} ms, upload_ms {s['upload_ms']:.3f}, index_build_ms {s['index_build_ms']:.3f}", flush=True)
    for rep in range(3):
        scan = np.ascontiguousarray(w.source + np.float32(1e-4 * rep))
        t0 = time.perf_counter(); e.setInputSource(scan); s = e.stats(); t1 = time.perf_counter()
        print(f"{cfg} setInputSource {len(scan)} points: host wall {1e3 * (t1 - t0):.3f} ms, upload_ms {s['upload_ms']:.3f}, index_build_ms {s['index_build_ms']:.3f}", flush=True)
