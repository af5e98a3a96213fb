"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the committed golden fixture, and — at BASELINE.json's full sizes — through
size-independent properties.  Tolerances (BASELINE.json north_star): final transform <= 1e-4 m / 1e-4 rad;
integer/index work (neighbour indices, correspondences, iteration counts) bit-exact; float32 squared
distances bit-exact; FP64 sums to 1e-10 relative (the reference's own reduction order is thread-schedule
dependent)."""
import numpy as np
import pytest

from direct_lidar_odometry_amd import clouds

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-4, 1e-4


@pytest.fixture(scope="module")
def ng(hip_lib):
    from direct_lidar_odometry_amd import nano_gicp
    return nano_gicp


def _pair(ng, orc, src, tgt, max_corr=None, **kw):
    g, o = ng.NanoGICP(), orc.OracleGICP()
    for e in (g, o):
        if max_corr is not None:
            e.setMaxCorrespondenceDistance(max_corr)
        for k, v in kw.items():
            getattr(e, k)(v)
        e.setInputSource(src)
        e.setInputTarget(tgt)
    return g, o


def _boundary_ties(orc, pts, k):
    """Points whose k-th and (k+1)-th neighbours are EXACTLY equidistant in float32: the reference keeps
    whichever its kd-tree visits first (impl/nanoflann_impl.hpp:184-211), which a grid cannot mirror
    (SURVEY.md §7 "Ties"); their covariance legitimately differs between the two searches."""
    if len(pts) <= k:
        return np.zeros(len(pts), bool)
    _, d2 = orc.OracleTree(pts).knn(pts, k + 1)
    return d2[:, k - 1] == d2[:, k]


def _share_covariances(g, o, orc, src, tgt, k):
    """Compute covariances on the GPU, check them against the oracle away from boundary ties, then give the
    oracle the GPU's set so that the registration loop is compared on identical inputs."""
    g.calculateSourceCovariances(); g.calculateTargetCovariances()
    o.calculateSourceCovariances(); o.calculateTargetCovariances()
    for pts, a, b in ((src, g.getSourceCovariances(), o.getSourceCovariances()), (tgt, g.getTargetCovariances(), o.getTargetCovariances())):
        ties = _boundary_ties(orc, pts, k)
        assert ties.mean() < 2e-3
        assert np.abs(a - b)[~ties].max() < 1e-9
    o.setSourceCovariances(g.getSourceCovariances()); o.setTargetCovariances(g.getTargetCovariances())


def _assert_pose_close(Tg, To, tt=TOL_T, tr=TOL_R):
    dt, dr = clouds.pose_error(Tg, To)
    assert dt <= tt and dr <= tr, (dt, dr)


# ------------------------------------------------------------------ golden fixture
def test_golden_fixture(ng, golden):
    src, tgt, p = golden["source"], golden["target"], golden["probes"]
    g = ng.NanoGICP(); g.setMaxCorrespondenceDistance(float(golden["max_corr_dist"]))
    g.setInputSource(src); g.setInputTarget(tgt)
    # neighbour search vs the REAL reference kd-tree's outputs
    for k, name in ((1, "ref_knn1"), (20, "ref_knn20")):
        idx, d2 = g.target_knn(src[p], k)
        assert np.array_equal(d2, golden[name + "_d2"]) and np.array_equal(idx, golden[name + "_idx"])
    g.calculateSourceCovariances(); g.calculateTargetCovariances()
    assert np.abs(g.getSourceCovariances()[p] - golden["cov_src_probes"]).max() < 1e-10
    assert np.abs(g.getTargetCovariances()[p] - golden["cov_tgt_probes"]).max() < 1e-10
    H, b, err = g.linearize(golden["guess"].astype(np.float64))
    assert np.array_equal(g.correspondences()[0], golden["corr"])
    assert abs(err - float(golden["err"])) <= 1e-10 * abs(err)
    assert np.abs(H - golden["H"]).max() <= 1e-10 * np.abs(H).max() and np.abs(b - golden["b"]).max() <= 1e-10 * np.abs(b).max()
    assert abs(g.compute_error(golden["T1"]) - float(golden["err_T1"])) <= 1e-10 * float(golden["err_T1"])
    g.align(golden["guess"])
    _assert_pose_close(g.getFinalTransformation(), golden["final_T"], 1e-6, 1e-6)
    assert g.nr_iterations_ == int(golden["nr_iterations"]) and g.converged_ == bool(golden["converged"])
    assert np.allclose(g.lm_trace(), golden["lm_trace"], rtol=1e-6, atol=1e-9)
    assert np.abs(g.getFinalHessian() - golden["final_hessian"]).max() <= 1e-9 * np.abs(golden["final_hessian"]).max()


# ------------------------------------------------------------------ neighbour search
@pytest.mark.parametrize("k", [1, 5, 10, 20, 32])
def test_knn_matches_oracle(ng, oracle_mod, k):
    w = clouds.scan_to_scan(10_000)
    g = ng.NanoGICP(); g.setInputTarget(w.target)
    rng = np.random.default_rng(k)
    far = (rng.normal(size=(300, 3)) * [40, 30, 8]).astype(np.float32)  # many queries outside the target's bounding box
    q = np.concatenate([w.source[::7], w.target[::50], far])
    gi, gd = g.target_knn(q, k)
    oi, od = oracle_mod.OracleTree(w.target).knn(q, k)
    assert np.array_equal(gd, od)  # float32 squared distances, bit-exact
    assert np.all(np.diff(gd, axis=1) >= 0)
    same = gi == oi
    assert np.all(gd[~same] == od[~same])  # an index may differ only on an exact distance tie
    assert same.mean() > 0.999


def test_knn_ties_duplicates_and_clamped_outliers(ng, oracle_mod):
    g3 = np.stack(np.meshgrid(np.arange(14), np.arange(10), np.arange(5), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.25
    pts = np.concatenate([g3, g3[:50], np.array([[500, 0, 0], [-300, 200, 50]], np.float32)])  # lattice + duplicates + far outliers
    q = np.concatenate([g3[::9] + np.float32(0.125), np.array([[480, 1, 1], [0, 0, 100]], np.float32)])
    g = ng.NanoGICP(); g.setInputTarget(pts)
    for k in (1, 8, 20):
        gi, gd = g.target_knn(q, k)
        oi, od = oracle_mod.OracleTree(pts).knn(q, k)
        assert np.array_equal(gd, od)
        # tie order is traversal dependent: compare as index SETS among strictly-closer-than-kth neighbours
        for r in range(len(q)):
            strict = gd[r] < gd[r, -1]
            assert set(gi[r][strict]) == set(oi[r][od[r] < od[r, -1]])


def test_tiny_and_degenerate_clouds(ng, oracle_mod):
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(25, 3)).astype(np.float32)
    g = ng.NanoGICP(); g.setCorrespondenceRandomness(20); g.setInputSource(pts); g.setInputTarget(pts + np.float32(0.01))
    g.calculateSourceCovariances()
    ties = _boundary_ties(oracle_mod, pts, 20)
    assert np.abs(g.getSourceCovariances() - oracle_mod.covariances(pts, 20))[~ties].max() < 1e-9
    flat = pts.copy(); flat[:, 2] = 1.0  # exactly planar cloud: zero extent in z
    g.setInputTarget(flat)
    gi, gd = g.target_knn(pts, 3)
    oi, od = oracle_mod.OracleTree(flat).knn(pts, 3)
    assert np.array_equal(gd, od) and np.array_equal(gi, oi)
    one = np.array([[1, 2, 3]], np.float32)
    g.setInputTarget(one)
    gi, gd = g.target_knn(pts[:5], 1)
    assert np.all(gi == 0)


def test_errors_are_reported_not_thrown(ng):
    g = ng.NanoGICP()
    with pytest.raises(ng.NgicpError) as e:
        g.align()
    assert e.value.code == -3  # no clouds: PCL's initCompute fails; here an explicit state error
    pts = np.random.default_rng(0).normal(size=(10, 3)).astype(np.float32)
    g.setInputSource(pts); g.setInputTarget(pts)
    with pytest.raises(ng.NgicpError) as e:
        g.calculateSourceCovariances()  # k = 20 > 10 points
    assert e.value.code == -4
    with pytest.raises(ng.NgicpError):
        g.setInputTarget(np.zeros((0, 3), np.float32))
    bad = pts.copy(); bad[3, 1] = np.nan
    with pytest.raises(ng.NgicpError):
        g.setInputTarget(bad)
    with pytest.raises(ng.NgicpError):
        g.setRegularizationMethod(9)
    with pytest.raises(ng.NgicpError):
        g.align()  # the rejected target left the slot empty
    g.setRegularizationMethod(ng.RegularizationMethod.PLANE)
    g.setCorrespondenceRandomness(5)
    g.setInputTarget(pts)
    g.align()  # still usable after errors


# ------------------------------------------------------------------ covariances
@pytest.mark.parametrize("reg", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("k", [10, 20])
def test_covariances_match_oracle(ng, oracle_mod, reg, k):
    w = clouds.scan_to_scan(10_000)
    g = ng.NanoGICP(); g.setCorrespondenceRandomness(k); g.setRegularizationMethod(reg)
    g.setInputSource(w.source); g.calculateSourceCovariances()
    a, b = g.getSourceCovariances(), oracle_mod.covariances(w.source, k, reg)
    assert a.shape == (10_000, 4, 4) and np.all(a[:, 3, :] == 0) and np.all(a[:, :, 3] == 0)
    ties = _boundary_ties(oracle_mod, w.source, k)
    assert ties.mean() < 2e-3 and np.abs(a - b)[~ties].max() < 1e-9
    assert np.abs(a - a.transpose(0, 2, 1)).max() == 0.0


@pytest.mark.parametrize("k", [3, 7, 15, 25, 32])
def test_covariances_k_off_the_list_sizes(ng, oracle_mod, k):
    """k that is not one of the compiled list sizes (10 / 20 / 32): the list is longer than k, the k-th best sits in either lane of
    the pair, and the neighbour sums stop inside a lane's half."""
    w = clouds.scan_to_scan(10_000)
    g = ng.NanoGICP(); g.setCorrespondenceRandomness(k); g.setInputSource(w.source); g.calculateSourceCovariances()
    a, b = g.getSourceCovariances(), oracle_mod.covariances(w.source, k, 3)
    ties = _boundary_ties(oracle_mod, w.source, k)
    assert ties.mean() < 5e-3 and np.abs(a - b)[~ties].max() < 1e-9


def test_set_get_covariances_roundtrip_and_reorder(ng):
    w = clouds.scan_to_scan(10_000)
    g = ng.NanoGICP(); g.setInputSource(w.source); g.setInputTarget(w.target)
    rng = np.random.default_rng(2)
    c = np.zeros((10_000, 4, 4)); m = rng.normal(size=(10_000, 3, 3)); c[:, :3, :3] = m @ m.transpose(0, 2, 1)
    g.setSourceCovariances(c); g.setTargetCovariances(c[::-1])
    assert np.array_equal(g.getSourceCovariances(), c) and np.array_equal(g.getTargetCovariances(), c[::-1])
    g.swapSourceAndTarget()
    assert np.array_equal(g.getTargetCovariances(), c) and g.sourceCovariancesSize() == 10_000
    g.setInputSource(w.source.copy())  # new cloud identity clears the cache (impl/nano_gicp_impl.hpp:128)
    assert g.sourceCovariancesSize() == 0


# ------------------------------------------------------------------ linearise / error
@pytest.mark.parametrize("max_corr", [None, 3.0, 1.0, 0.3])
def test_linearize_and_error_match_oracle(ng, oracle_mod, max_corr):
    w = clouds.scan_to_scan(10_000)
    g, o = _pair(ng, oracle_mod, w.source, w.target, max_corr)
    g.calculateSourceCovariances(); g.calculateTargetCovariances()
    o.setSourceCovariances(g.getSourceCovariances()); o.setTargetCovariances(g.getTargetCovariances())
    for T in (np.eye(4), w.gt, clouds.make_pose((1.0, -2.0, 0.3), (3, -2, 25))):
        Hg, bg, eg = g.linearize(T)
        Ho, bo, eo = o.linearize(T)
        cg, sg = g.correspondences(); co, so = o.correspondences()
        assert np.array_equal(cg, co)  # bit-exact correspondence indices (incl. -1 for gated-out points)
        assert np.array_equal(sg[cg >= 0], so[co >= 0])
        assert abs(eg - eo) <= 1e-10 * abs(eo)
        assert np.abs(Hg - Ho).max() <= 1e-10 * np.abs(Ho).max() and np.abs(bg - bo).max() <= 1e-10 * np.abs(bo).max()
        assert np.array_equal(Hg, Hg.T)
        T2 = clouds.make_pose((0.02, 0.01, -0.01), (0.2, 0.1, -0.3)) @ T
        assert abs(g.compute_error(T2) - o.compute_error(T2)) <= 1e-10 * o.compute_error(T2)


@pytest.mark.parametrize("shape", ["cube", "plane", "lines", "clumps"])
def test_correspondences_exact_on_adversarial_shapes(ng, oracle_mod, shape):
    """The search is exact by construction (total order, conservative bounds); this hammers its corner cases with target
    geometries unlike a LiDAR scan: uniform volume, one dense plane, dense lines along x / y / z (long x-sorted rows, rows
    whose points share one x), tight clumps (cells with hundreds of points next to empty space), at several poses and gates.
    1-NN ties between equidistant points are excluded from the index comparison (kd-tree visiting order)."""
    rng = np.random.default_rng({"cube": 1, "plane": 2, "lines": 3, "clumps": 4}[shape])
    n = 6000
    if shape == "cube":
        tgt = rng.uniform(-3, 3, (n, 3))
    elif shape == "plane":
        tgt = np.c_[rng.uniform(-4, 4, (n, 2)), 0.002 * rng.standard_normal(n)]
    elif shape == "lines":
        t = rng.uniform(-4, 4, n); k = rng.integers(0, 3, n); off = rng.integers(-3, 4, (n, 2)) * 0.5
        tgt = np.zeros((n, 3))
        for ax in range(3):
            m = k == ax
            tgt[m, ax] = t[m]
            tgt[np.ix_(m, [a for a in range(3) if a != ax])] = off[m]
        tgt += 0.001 * rng.standard_normal((n, 3))
    else:
        centres = rng.uniform(-3, 3, (12, 3))
        tgt = centres[rng.integers(0, 12, n)] + 0.01 * rng.standard_normal((n, 3))
    tgt = tgt.astype(np.float32)
    src = (tgt[rng.permutation(n)[:3000]] + rng.normal(0, 0.05, (3000, 3))).astype(np.float32)
    src = np.r_[src, rng.uniform(-5, 5, (500, 3)).astype(np.float32)]  # and some queries far from everything
    tree = oracle_mod.OracleTree(tgt)
    for gate in (None, 2.0, 0.4):
        g, o = _pair(ng, oracle_mod, src, tgt, gate)
        covs = np.tile(np.eye(4)[None] * np.array([1, 1, 1, 0])[None, :, None], (1, 1, 1))
        cs = np.repeat(covs, len(src), 0); ct = np.repeat(covs, len(tgt), 0)
        for e in (g, o):
            e.setSourceCovariances(cs); e.setTargetCovariances(ct)
        for T in (np.eye(4), clouds.make_pose((0.3, -0.2, 0.1), (2, -1, 5)), clouds.make_pose((0.31, -0.2, 0.1), (2, -1, 5.1))):
            g.linearize(T); o.linearize(T)
            cg, sg = g.correspondences(); co, so = o.correspondences()
            q = clouds.transform_points(T, src)
            _, d2 = tree.knn(q, 2)
            tie = (d2[:, 1] - d2[:, 0]) <= 1e-5 * np.maximum(d2[:, 1], 1e-12)  # (numpy's transform may round differently)
            assert np.array_equal(cg >= 0, co >= 0)
            assert np.array_equal(cg[~tie], co[~tie])
            assert np.array_equal(sg[cg >= 0], so[co >= 0])  # the distances agree even where a tie picks another point


# ------------------------------------------------------------------ full alignment
CASES = {
    "dlo_s2s": dict(setMaximumIterations=32, setTransformationEpsilon=0.01, setCorrespondenceRandomness=10),   # cfg/params.yaml:54-62
    "dlo_s2m": dict(setMaximumIterations=32, setTransformationEpsilon=0.01, setCorrespondenceRandomness=20),   # cfg/params.yaml:63-71
    "defaults": dict(),
    "fixed20": dict(setMaximumIterations=20, setTransformationEpsilon=1e-12, setRotationEpsilon=1e-12),
    "gauss_newton": dict(setOptimizer=0, setMaximumIterations=15),
    "one_iteration": dict(setMaximumIterations=1),
}


@pytest.mark.parametrize("case", list(CASES))
def test_align_matches_oracle_scan_to_scan(ng, oracle_mod, case):
    w = clouds.scan_to_scan(10_000)
    corr = {"dlo_s2s": 1.0, "dlo_s2m": 0.5}.get(case, None if case == "defaults" else 1.0)
    g, o = _pair(ng, oracle_mod, w.source, w.target, corr, **CASES[case])
    g.align(); o.align()  # end to end, each side estimating its own covariances inside align()
    _assert_pose_close(g.getFinalTransformation(), o.getFinalTransformation())
    # strict comparison of the registration loop on identical covariances (boundary ties excluded above)
    g, o = _pair(ng, oracle_mod, w.source, w.target, corr, **CASES[case])
    _share_covariances(g, o, oracle_mod, w.source, w.target, CASES[case].get("setCorrespondenceRandomness", 20))
    g.align(); o.align()
    _assert_pose_close(g.getFinalTransformation(), o.getFinalTransformation(), 1e-6, 1e-6)
    tg, to = g.lm_trace(), o.lm_trace()
    if case == "fixed20":
        # with eps = 1e-12 the loop only ends on max_iterations or when LM gives up after 10 rejected trials
        # at the noise floor (rho ~ 0/0); WHICH iteration that happens in is rounding noise, so only the
        # common prefix of the trace and the final pose are comparable
        m = min(len(tg), len(to), 4)
        assert np.array_equal(tg[:m, [0, 1, 7]], to[:m, [0, 1, 7]]) and np.allclose(tg[:m, 2:4], to[:m, 2:4], rtol=1e-6)
        return
    assert g.nr_iterations_ == o.nr_iterations and g.converged_ == o.converged
    assert tg.shape == to.shape
    if len(to):
        assert np.array_equal(tg[:, [0, 1, 7]], to[:, [0, 1, 7]])          # same accept/reject sequence
        assert np.allclose(tg[:, 2:4], to[:, 2:4], rtol=1e-6)               # y0, yi
    assert np.abs(g.getFinalHessian() - o.getFinalHessian()).max() <= 1e-6 * np.abs(o.getFinalHessian()).max()


@pytest.mark.parametrize("case", ["dlo_s2s", "dlo_s2m", "gauss_newton"])
def test_correspondences_after_align_are_exact(ng, oracle_mod, case):
    """Every pass after the first starts its search warm (previous correspondence) and groups are launched in a measured-cost
    order: neither may change a single nearest neighbour.  After align() the reference's correspondences_ are those of its
    last linearisation (impl/nano_gicp_impl.hpp:174-211 called from :219); compare them index for index."""
    w = clouds.scan_to_scan(10_000)
    corr = {"dlo_s2s": 1.0, "dlo_s2m": 0.5}.get(case, 1.0)
    g, o = _pair(ng, oracle_mod, w.source, w.target, corr, **CASES[case])
    _share_covariances(g, o, oracle_mod, w.source, w.target, CASES[case].get("setCorrespondenceRandomness", 20))
    g.align(); o.align()
    assert g.nr_iterations_ == o.nr_iterations and g.nr_iterations_ >= 2  # several warm-started passes happened
    cg, _ = g.correspondences(); co, _ = o.correspondences()
    assert np.array_equal(cg, co)


def test_align_scan_to_submap_with_supplied_covariances(ng, oracle_mod):
    """Config-3 shape at reduced size: per-keyframe world-frame covariances concatenated (odom.cc:1318-1325,833)."""
    w = clouds.scan_to_submap(6_000, 3)
    covs = np.concatenate([oracle_mod.covariances(k, 20) for k in np.split(w.target, np.cumsum(w.keyframe_sizes)[:-1])])
    g, o = _pair(ng, oracle_mod, w.source, w.target, w.max_corr_dist, setMaximumIterations=32, setTransformationEpsilon=0.01)
    g.setTargetCovariances(covs); o.setTargetCovariances(covs)
    g.align(w.guess); o.align(w.guess)
    _assert_pose_close(g.getFinalTransformation(), o.getFinalTransformation())
    assert g.nr_iterations_ == o.nr_iterations and g.converged_ == o.converged
    assert g.targetCovariancesSize() == len(w.target)  # supplied set kept, not recomputed


def test_lm_rejection_path_matches_oracle(ng, oracle_mod):
    """A poor guess + huge initial lambda factor forces rejected trials (impl/lsq_registration_impl.hpp:191-199)."""
    w = clouds.scan_to_scan(10_000)
    guess = clouds.make_pose((1.5, -1.0, 0.2), (2, -3, 12)).astype(np.float32)
    g, o = _pair(ng, oracle_mod, w.source, w.target, 2.0, setMaximumIterations=12, setInitialLambdaFactor=1e-15)
    g.align(guess); o.align(guess)
    tg, to = g.lm_trace(), o.lm_trace()
    assert tg.shape == to.shape and np.array_equal(tg[:, [0, 1, 7]], to[:, [0, 1, 7]])
    _assert_pose_close(g.getFinalTransformation(), o.getFinalTransformation())
    assert g.nr_iterations_ == o.nr_iterations and g.converged_ == o.converged


def test_zero_correspondences(ng, oracle_mod):
    w = clouds.scan_to_scan(10_000)
    g, o = _pair(ng, oracle_mod, w.source[:500], w.target[:500] + np.float32(80), 1e-6)
    g.align(); o.align()
    assert np.array_equal(g.getFinalTransformation(), o.getFinalTransformation())
    assert g.converged_ == o.converged and g.nr_iterations_ == o.nr_iterations


def test_aligned_output_cloud(ng):
    w = clouds.scan_to_scan(10_000)
    g = ng.NanoGICP(); g.setMaxCorrespondenceDistance(1.0); g.setInputSource(clouds.to_xyzi(w.source)); g.setInputTarget(clouds.to_xyzi(w.target))
    out = g.align(want_aligned=True)
    T = g.getFinalTransformation()
    ref = w.source @ T[:3, :3].T + T[:3, 3]
    assert out.shape == (10_000, 3) and np.abs(out - ref).max() < 1e-5  # float32 transform, original point order


# ------------------------------------------------------------------ DLO call sequence (boundary contract)
def test_dlo_call_sequence_two_instances(ng, oracle_mod):
    sc = clouds.make_scene()
    scans = [clouds.vlp16(sc, clouds.make_pose((0.3 * i, 0.1 * i, 0.0), (0, 0, 2.0 * i)), noise_seed=10 + i, cols=375) for i in range(3)]

    def run(mk):
        s2s, s2m = mk(), mk()
        for e, k, d in ((s2s, 10, 1.0), (s2m, 20, 0.5)):  # odom.cc:100-114
            e.setCorrespondenceRandomness(k); e.setMaxCorrespondenceDistance(d)
            e.setMaximumIterations(32); e.setTransformationEpsilon(0.01)
        out = []
        s2s.setInputTarget(scans[0]); s2s.calculateTargetCovariances()                       # odom.cc:479-480
        s2s.setInputSource(scans[0]); s2s.calculateSourceCovariances()                       # odom.cc:498-499
        kf_covs = s2s.getSourceCovariances()                                                 # odom.cc:500
        submap, submap_covs = scans[0], kf_covs
        T_prev = np.eye(4, dtype=np.float32)
        for i in (1, 2):
            s2s.setInputSource(scans[i])                                                     # odom.cc:519
            s2m.registerInputSource(scans[i])                                                # odom.cc:522
            s2m.shareSourceIndexFrom(s2s); s2m.clearSourceCovariances()                      # odom.cc:525-526
            s2s.align()                                                                      # odom.cc:805
            T_s2s = s2s.getFinalTransformation().copy()
            s2m.copySourceCovariancesFrom(s2s)                                               # odom.cc:815
            s2s.swapSourceAndTarget()                                                        # odom.cc:818
            if i == 1:
                s2m.setInputTarget(submap); s2m.setTargetCovariances(submap_covs)            # odom.cc:830-833
            guess = (T_prev.astype(np.float64) @ T_s2s.astype(np.float64)).astype(np.float32)
            s2m.align(guess)                                                                 # odom.cc:837
            T_prev = s2m.getFinalTransformation().copy()
            out.append((T_s2s, T_prev.copy(), s2s.nr_iterations_ if hasattr(s2s, "nr_iterations_") else s2s.nr_iterations,
                        s2m.nr_iterations_ if hasattr(s2m, "nr_iterations_") else s2m.nr_iterations))
        return out, kf_covs

    a, ca = run(ng.NanoGICP)
    b, cb = run(oracle_mod.OracleGICP)
    ties = _boundary_ties(oracle_mod, scans[0], 10)
    assert np.abs(ca - cb)[~ties].max() < 1e-9
    for (Ta1, Ta2, na1, na2), (Tb1, Tb2, nb1, nb2) in zip(a, b):
        _assert_pose_close(Ta1, Tb1); _assert_pose_close(Ta2, Tb2)
        assert (na1, na2) == (nb1, nb2)


# ------------------------------------------------------------------ point-sharded stepping on one GPU
def test_sharded_stepping_equals_single_align(ng):
    import torch
    from direct_lidar_odometry_amd import sharding as sh
    w = clouds.scan_to_scan(10_000)
    full = ng.NanoGICP(); full.setMaxCorrespondenceDistance(1.0); full.setInputSource(w.source); full.setInputTarget(w.target)
    full.align()
    cs, ct = full.getSourceCovariances(), full.getTargetCovariances()
    world = 2
    engines = []
    for r in range(world):
        lo, hi = sh.shard_bounds(len(w.source), world, r)
        e = ng.NanoGICP(); e.setMaxCorrespondenceDistance(1.0)
        e.setInputSource(w.source[lo:hi]); e.setInputTarget(w.target)
        e.setSourceCovariances(cs[lo:hi]); e.setTargetCovariances(ct)
        e.sharded_begin()
        engines.append(e)
    dev = torch.device("cuda:0")
    bufs = [torch.zeros(sh.SUMS_LEN, dtype=torch.float64, device=dev) for _ in range(world)]
    for _ in range(200):
        for e, bfr in zip(engines, bufs):
            e.sharded_pass(bfr.data_ptr())
        torch.cuda.synchronize()
        total = bufs[0] + bufs[1]  # stands in for the RCCL all-reduce
        torch.cuda.synchronize()
        done = [e.sharded_step(total.data_ptr()) for e in engines]
        assert len(set(done)) == 1
        if done[0]:
            break
    Ts = [e.sharded_finish() for e in engines]
    assert np.array_equal(Ts[0], Ts[1])  # every rank ends on the identical pose without a broadcast
    _assert_pose_close(Ts[0], full.getFinalTransformation(), 1e-6, 1e-6)
    assert engines[0].nr_iterations_ == full.nr_iterations_ and engines[0].converged_ == full.converged_


# ------------------------------------------------------------------ full-size properties (BASELINE configs 2/3)
def test_full_size_properties(ng):
    w = clouds.scan_to_submap(100_000, 5)
    runs = []
    for vox in (0.0, 0.2, 0.35):  # automatic voxel, then two fixed grids
        g = ng.NanoGICP(); g.setTuning(vox); g.setMaxCorrespondenceDistance(w.max_corr_dist)
        g.setMaximumIterations(20); g.setTransformationEpsilon(1e-12); g.setRotationEpsilon(1e-12)
        g.setInputTarget(w.target); g.setInputSource(w.source)
        if not runs:
            g.calculateSourceCovariances(); g.calculateTargetCovariances()
            covs = (g.getSourceCovariances(), g.getTargetCovariances())
            idx, d2 = g.target_knn(w.source[::500], 20)
            bd = np.stack([np.sort(np.sum((w.target - q) ** 2, axis=1, dtype=np.float32))[:20] for q in w.source[::500]])
            assert np.allclose(d2, bd, rtol=2e-6, atol=1e-9)  # exact search at 500k points vs brute force (numpy sums in another order)
            assert np.all(np.diff(d2, axis=1) >= 0)
        else:
            g.setSourceCovariances(covs[0]); g.setTargetCovariances(covs[1])
        g.align(w.guess)
        T1 = g.getFinalTransformation().copy(); tr1 = g.lm_trace().copy()
        g.align(w.guess)
        assert np.array_equal(T1, g.getFinalTransformation()) and np.array_equal(tr1, g.lm_trace())  # run-to-run bitwise reproducible
        runs.append((T1, tr1, g.nr_iterations_))
        s = g.stats()
        assert s["passes"] == s["lm_trials"] + 1 and 0.5 < s["valid_fraction"] <= 1.0
    # the grid resolution is a pure performance knob: exact search => identical results
    for T, tr, n in runs[1:]:
        assert np.array_equal(T, runs[0][0]) and np.array_equal(tr[:, [0, 1, 7]], runs[0][1][:, [0, 1, 7]]) and n == runs[0][2]
        # a different grid changes the summation order: the first iterations agree to rounding; later the 1e-12 pose
        # differences (the 6x6 solve amplifies rounding by cond(H)) can flip a nearest-neighbour or gate decision that sits
        # exactly on its boundary, which moves y0 / yi by ~1e-6 relative while the float32 result stays identical
        assert np.allclose(tr[:3, [2, 3]], runs[0][1][:3, [2, 3]], rtol=1e-9) and np.allclose(tr[:, [2, 3]], runs[0][1][:, [2, 3]], rtol=1e-4)
        assert np.allclose(tr[:, [5, 6]], runs[0][1][:, [5, 6]], rtol=1e-2)
        assert np.allclose(tr[:, 4], runs[0][1][:, 4], atol=1e-2)


@pytest.mark.parametrize("switch", ["NGICP_PERSIST", "NGICP_HEAD", "NGICP_CELL_BOXES"])
@pytest.mark.parametrize("case", ["dlo_s2s", "fixed20", "gauss_newton", "one_iteration"])
def test_persistent_kernel_equals_one_launch_per_pass(ng, monkeypatch, case, switch):
    """The two ways of running an iteration without a solver launch, against the default (one pass launch + one solver launch per
    iteration).  NGICP_PERSIST=1: ONE launch per alignment - the blocks keep their groups, meet after every pass, the last one steps the
    optimiser.  NGICP_HEAD=1: one launch per iteration - the blocks add their rows up per subset, and every block of the next launch steps
    the optimiser itself at its head.  The same sums in the same order, so everything is bit-identical - pose, Hessian, trace,
    correspondences.  A 100k-point source has more groups (866) than the persistent grid has blocks (768): some blocks take two groups
    per pass; 10k points leave most of the grid's blocks without a group of their own.  NGICP_CELL_BOXES=1 (per-cell (y,z) extents that
    cut and prune the ring-1 pairs) rides along: an exact search finds the same neighbours, so it is bit-identical too."""
    for w in (clouds.scan_to_scan(10_000), clouds.scan_to_submap(100_000, 5)):
        if len(w.source) > 50_000 and case not in ("fixed20", "one_iteration"):
            continue
        out = []
        for persist in ("0", "1"):
            monkeypatch.setenv(switch, persist)  # (read when the handle is created)
            g = ng.NanoGICP(); g.setMaxCorrespondenceDistance(w.max_corr_dist)
            for k, v in CASES[case].items():
                getattr(g, k)(v)
            g.setInputTarget(w.target); g.setInputSource(w.source)
            runs = []
            for _ in range(3):  # (the second and third alignment launch in the order the one before left behind)
                g.align(w.guess)
                runs.append((g.getFinalTransformation().copy(), g.getFinalHessian().copy(), g.lm_trace().copy(), g.nr_iterations_, g.converged_, g.stats()["passes"]))
            corr = g.correspondences()
            out.append((runs, corr))
        monkeypatch.delenv(switch)
        for (a, b) in zip(out[0][0], out[1][0]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3:] == b[3:]
        assert np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])
        assert out[0][0][0][5] >= 1


# ------------------------------------------------------------------ the small FP64 routines, directly on the device
def test_device_math_matches_oracle(ng, oracle_mod):
    """so3_exp (both branches: the Taylor series for theta^2 < 1e-10 and the sin / cos form, gicp/so3.hpp:99-118), the 6x6 LDLT
    solve, the symmetric 3x3 eigen-decomposition and inverse, evaluated by the HIP build of csrc/ngicp_math.h and compared with
    the oracle's own restatements (and with numpy where a closed form exists)."""
    g = ng.NanoGICP()
    rng = np.random.default_rng(11)
    # so3_exp: tiny rotations (Taylor branch), the branch boundary, ordinary and large rotations
    w = np.concatenate([rng.normal(size=(50, 3)) * 1e-7, rng.normal(size=(50, 3)) * 3e-6, np.array([[1e-5, 0, 0], [0, 1.0000001e-5, 0], [0, 0, 0]]),
                        rng.normal(size=(100, 3)) * 0.01, rng.normal(size=(100, 3)), rng.normal(size=(20, 3)) * 3.0])
    R = g.mathSelftest(0, w).reshape(-1, 3, 3)
    Ro = np.stack([oracle_mod.so3_exp(x) for x in w])
    assert np.abs(R - Ro).max() <= 2e-15  # same formula, same order: a few roundings apart at most (device libm sqrt / sin / cos)
    assert (np.sum(w ** 2, axis=1) < 1e-10).sum() >= 50 and (np.sum(w ** 2, axis=1) >= 1e-10).sum() >= 100  # both branches exercised
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-14
    # 6x6 LDLT: well-conditioned SPD, GICP-like (badly scaled) and a semi-definite one with a zero pivot
    A = []
    for i in range(60):
        m = rng.normal(size=(6, 6)); a = m @ m.T + (1e-3 if i % 3 else 1e3) * np.eye(6)
        if i % 3 == 2:
            s = np.diag([1e6, 1e6, 1e6, 1.0, 1.0, 1.0]); a = s @ a @ s
        A.append(a)
    semi = np.diag([2.0, 0.0, 3.0, 1.0, 5.0, 4.0]); A.append(semi)
    A = np.stack(A); rhs = rng.normal(size=(len(A), 6))
    x = g.mathSelftest(1, np.concatenate([A.reshape(len(A), 36), rhs], axis=1))
    xo = np.stack([oracle_mod.ldlt6_solve(a, r) for a, r in zip(A, rhs)])
    assert np.allclose(x, xo, rtol=1e-9, atol=1e-12)
    assert np.allclose(np.einsum("nij,nj->ni", A[:-1], x[:-1]), rhs[:-1], rtol=1e-6, atol=1e-6)
    assert x[-1, 1] == 0.0  # a zero pivot contributes zero, like Eigen's LDLT::solve
    # symmetric 3x3: covariance-like matrices incl. nearly planar / nearly linear ones
    C = []
    for i in range(200):
        p = rng.normal(size=(20, 3)) * np.array([1.0, 10.0 ** -(i % 5), 10.0 ** -(i % 7)]); p -= p.mean(0); c = p.T @ p / 20
        q, _ = np.linalg.qr(rng.normal(size=(3, 3))); c = q @ c @ q.T
        C.append([c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]])
    C = np.array(C)
    ev = g.mathSelftest(2, C)
    full = np.stack([[[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]] for c in C])
    wv, V = ev[:, :3], ev[:, 3:].reshape(-1, 3, 3)
    assert np.allclose(np.sort(wv, axis=1), np.linalg.eigvalsh(full), rtol=1e-9, atol=1e-15)
    assert np.abs(np.einsum("nij,nj,nkj->nik", V, wv, V) - full).max() < 1e-12  # V diag(w) V^T reproduces the matrix
    for c, (w3, V3) in zip(C[:40], zip(wv, V)):
        wo, Vo = oracle_mod.eig3_sym(np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]]))
        i3, io = np.argsort(-w3), np.argsort(-wo)  # (the two return their eigenpairs in different orders)
        assert np.allclose(w3[i3], wo[io], rtol=1e-12, atol=1e-18)
        if np.min(np.abs(np.diff(w3[i3]))) > 1e-6 * np.abs(w3).max():  # eigenvectors are only defined for separated eigenvalues
            assert np.allclose(np.abs(V3[:, i3]), np.abs(Vo[:, io]), atol=1e-7)
    Ci = C.copy(); Ci[:, [0, 3, 5]] += 1e-3
    inv = g.mathSelftest(3, Ci)
    fulli = np.stack([[[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]] for c in Ci])
    got = np.stack([[[m[0], m[1], m[2]], [m[1], m[3], m[4]], [m[2], m[4], m[5]]] for m in inv])
    assert np.allclose(got, np.linalg.inv(fulli), rtol=1e-8, atol=0)


def test_long_thin_cloud_with_a_small_user_voxel(ng, oracle_mod):
    """the pass kernel packs a listed row as y | z << 16, so a grid may not have more than 65535 rows in y (32767 in z):
    a 4 km long, 1 m wide cloud with a 5 cm user voxel would have 80 000 - make_grid grows the voxel instead.  The alignment must
    still equal the oracle's (gate wide enough that rows beyond ring 1 are listed)."""
    rng = np.random.default_rng(5)
    n = 20_000
    tgt = np.c_[rng.uniform(-0.5, 0.5, n), rng.uniform(0.0, 4000.0, n), rng.uniform(-0.5, 0.5, n)].astype(np.float32)
    T = clouds.make_pose((0.03, -0.05, 0.02), (0.0, 0.0, 0.0003))  # (4 km from the origin: 0.0003 deg are 2 cm)
    src = clouds.transform_points(np.linalg.inv(T), tgt[::2] + rng.normal(0, 0.005, (n // 2, 3)).astype(np.float32))
    g, o = ng.NanoGICP(), oracle_mod.OracleGICP()
    g.setTuning(0.05)
    for e in (g, o):
        e.setCorrespondenceRandomness(10); e.setMaxCorrespondenceDistance(0.6)
        e.setInputSource(src); e.setInputTarget(tgt)
    g.calculateSourceCovariances(); g.calculateTargetCovariances()
    s = g.stats()
    o.setSourceCovariances(g.getSourceCovariances()); o.setTargetCovariances(g.getTargetCovariances())
    Hg, bg, eg = g.linearize(np.eye(4)); Ho, bo, eo = o.linearize(np.eye(4))
    cg, sg = g.correspondences(); co, so = o.correspondences()
    assert g.stats()["grid_dims"][1] < 65536 and g.stats()["voxel_size"] > 0.05
    assert np.array_equal(cg, co) and np.array_equal(sg[cg >= 0], so[co >= 0]) and (cg >= 0).mean() > 0.5
    assert abs(eg - eo) <= 1e-9 * abs(eo) and np.abs(Hg - Ho).max() <= 1e-9 * np.abs(Ho).max()
    g.align(); o.align()
    _assert_pose_close(g.getFinalTransformation(), o.getFinalTransformation())
    assert g.nr_iterations_ == o.nr_iterations
