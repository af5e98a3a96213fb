"""GPU tests (-m gpu): several handles driven from several host threads at once.  DLO's node runs under ros::AsyncSpinner(0)
(src/dlo/odom_node.cc): a handle has one caller at a time, but different handles are called from different threads, and the
engine's process-wide pools (index objects, covariance buffers) and the event fences between the handles' streams are shared by
all of them.  Results must be those of the same calls made one after the other, bit for bit."""
import threading

import numpy as np
import pytest

from direct_lidar_odometry_amd import clouds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ng(hip_lib):
    from direct_lidar_odometry_amd import nano_gicp
    return nano_gicp


def _frames(ng, w, scans):
    """The per-frame sequence of odom.cc:498-526, 803-840 on a private pair of handles; returns every pose and the last covariances."""
    s2s, s2m = ng.NanoGICP(), ng.NanoGICP()
    for e, k, d in ((s2s, 10, 1.0), (s2m, 20, 0.5)):
        e.setCorrespondenceRandomness(k); e.setMaxCorrespondenceDistance(d); e.setMaximumIterations(32); e.setTransformationEpsilon(0.01)
    s2m.setInputTarget(w.target); s2m.calculateTargetCovariances()
    out = []
    prev = None
    for scan in scans:
        s2s.setInputSource(scan); s2s.calculateSourceCovariances()
        if prev is not None:
            s2s.align()
            out.append(s2s.getFinalTransformation().copy())
        s2m.registerInputSource(scan); s2m.shareSourceIndexFrom(s2s); s2m.copySourceCovariancesFrom(s2s)
        s2m.align(w.guess)
        out.append(s2m.getFinalTransformation().copy())
        s2s.swapSourceAndTarget()   # odom.cc:818: the scan becomes the next frame's target
        prev = scan
    covs = s2m.getSourceCovariances().copy()
    s2s.close(); s2m.close()
    return out, covs


@pytest.mark.parametrize("persist", ["0", "1"])
def test_handles_on_concurrent_threads_give_the_serial_results(ng, monkeypatch, persist):
    """persist = 1: the handles ask for the persistent registration kernel (NGICP_PERSIST, read at ngicp_create).  One alignment per
    device at a time takes that route (its grid fills the device and its blocks wait for each other); a handle that finds it taken runs
    one launch per pass beside it.  Either route gives the same bits."""
    w = clouds.scan_to_submap(20_000, 3)
    n_threads, n_frames = 4, 5
    scans = [[np.ascontiguousarray(w.source + np.float32(1e-3 * (5 * t + i))) for i in range(n_frames)] for t in range(n_threads)]
    monkeypatch.setenv("NGICP_PERSIST", "0")
    serial = [_frames(ng, w, scans[t]) for t in range(n_threads)]
    monkeypatch.setenv("NGICP_PERSIST", persist)
    results, errors = [None] * n_threads, []

    def work(t):
        try:
            results[t] = _frames(ng, w, scans[t])
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((t, repr(e)))

    for _ in range(2):  # twice: the second round runs on recycled pool objects
        threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=120)
        assert not errors, errors
        assert all(not th.is_alive() for th in threads)
        for t in range(n_threads):
            poses, covs = results[t]
            assert len(poses) == len(serial[t][0])
            for a, b in zip(poses, serial[t][0]):
                assert np.array_equal(a, b)
            assert np.array_equal(covs, serial[t][1])


def test_yielding_host_wait_gives_the_same_alignment(ng=None):
    """ngicp_set_host_wait(1): the calling thread yields its core between polls of the solver's progress word instead of spinning
    on it (DLO's callbacks share an AsyncSpinner's threads, src/dlo/odom_node.cc:27).  How the host waits cannot change what the
    device computes."""
    import numpy as np
    from direct_lidar_odometry_amd import clouds, nano_gicp
    w = clouds.scan_to_scan(20_000)
    out = []
    for mode in (0, 1):
        g = nano_gicp.NanoGICP(); g.setHostWaitMode(mode); g.setMaxCorrespondenceDistance(1.0)
        g.setInputSource(w.source); g.setInputTarget(w.target); g.align()
        out.append((g.getFinalTransformation().copy(), g.nr_iterations_, g.stats()["host_wait_spins"]))
        g.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert out[0][2] >= 0 and out[1][2] >= 0
    with __import__("pytest").raises(nano_gicp.NgicpError):
        g2 = nano_gicp.NanoGICP(); g2.setHostWaitMode(7)
