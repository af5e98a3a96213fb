"""The C++ host side: the header-only nano_gicp::NanoGICP shim (include/nano_gicp/nano_gicp.hpp) compiled with
g++ against the C ABI, replaying DLO's own call sequence (tests/cpp/replay_odom.cpp mirrors src/dlo/odom.cc)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "replay_odom")


def _build(hip_lib):
    libdir = os.path.join(ROOT, "direct_lidar_odometry_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "replay_odom.cpp"),
           "-o", BIN, "-L" + libdir, "-lngicp_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return BIN


def test_shim_compiles_and_fails_loudly_without_gpu(hip_lib, tmp_path):
    import torch
    exe = _build(hip_lib)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    f = tmp_path / "s.bin"
    np.zeros((30, 3), np.float32).tofile(f)
    res = subprocess.run([exe, "2", str(f), str(f)], capture_output=True, text=True)
    assert res.returncode == 3 and "cannot create the GPU engine" in res.stderr  # no silent CPU path


@pytest.mark.gpu
def test_cpp_shim_replays_dlo_sequence_like_the_oracle(hip_lib, oracle_mod, tmp_path):
    from direct_lidar_odometry_amd import clouds
    exe = _build(hip_lib)
    sc = clouds.make_scene()
    scans = [clouds.vlp16(sc, clouds.make_pose((0.3 * i, 0.1 * i, 0.0), (0, 0, 2.0 * i)), noise_seed=10 + i, cols=375) for i in range(3)]
    paths = []
    for i, s in enumerate(scans):
        p = tmp_path / f"scan{i}.bin"
        s.tofile(p)
        paths.append(str(p))
    res = subprocess.run([exe, str(len(scans)), *paths], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = {"s2s": [], "s2m": [], "aligned": [], "covs": []}
    for line in res.stdout.splitlines():
        tag, *vals = line.split()
        got[tag].append([float(v) for v in vals])

    # the same sequence on the CPU oracle
    O = oracle_mod.OracleGICP
    s2s, s2m = O(), O()
    for e, k, d in ((s2s, 10, 1.0), (s2m, 20, 0.5)):
        e.setCorrespondenceRandomness(k); e.setMaxCorrespondenceDistance(d); e.setMaximumIterations(32); e.setTransformationEpsilon(0.01)
    s2s.setInputTarget(scans[0]); s2s.calculateTargetCovariances()
    s2s.setInputSource(scans[0]); s2s.calculateSourceCovariances()
    kf = s2s.getSourceCovariances()
    assert int(got["covs"][0][0]) == len(kf)
    assert abs(got["covs"][0][1] - kf[0][0, 0]) < 1e-9 and abs(got["covs"][0][2] - kf[0][1, 2]) < 1e-9
    T_prev = np.eye(4, dtype=np.float32)
    for i in (1, 2):
        s2s.setInputSource(scans[i]); s2m.registerInputSource(scans[i]); s2m.shareSourceIndexFrom(s2s); s2m.clearSourceCovariances()
        s2s.align(); T1 = s2s.getFinalTransformation().copy()
        s2m.copySourceCovariancesFrom(s2s); s2s.swapSourceAndTarget()
        if i == 1:
            s2m.setInputTarget(scans[0]); s2m.setTargetCovariances(kf)
        guess = (T_prev.astype(np.float32) @ T1.astype(np.float32)).astype(np.float32)  # float product, as Eigen::Matrix4f
        s2m.align(guess); T_prev = s2m.getFinalTransformation().copy()
        for tag, T, e in (("s2s", T1, s2s), ("s2m", T_prev, s2m)):
            row = got[tag][i - 1]
            Tg = np.array(row[:16]).reshape(4, 4)
            dt, dr = clouds.pose_error(Tg, T)
            assert dt <= 1e-4 and dr <= 1e-4, (tag, i, dt, dr)
            assert int(row[16]) == e.nr_iterations and bool(int(row[17])) == e.converged
        a = got["aligned"][i - 1]
        ref = T_prev[:3, :3] @ scans[i][7] + T_prev[:3, 3]
        assert int(a[0]) == len(scans[i]) and np.abs(np.array(a[1:4]) - ref).max() < 1e-3 and a[4] == 1.0
