"""GPU tests (-m gpu) of the rows SURVEY.md §8f lists next to the registration path:
  f1  device-resident keyframe store + submap assembly   (src/dlo/odom.cc:1166-1174, 1240-1331, 827-834)
  f3  rigid transform of clouds                          (impl/lsq_registration_impl.hpp:114, odom.cc:484, 971-974)
Checker: the CPU oracle fed the host-concatenated cloud + covariances (f1) / the oracle's restatement of
pcl::transformPointCloud (f3; PCL's source is not under /root/reference: parity unpinned, see ngicp_oracle.cpp)."""
import numpy as np
import pytest

from direct_lidar_odometry_amd import clouds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ng(hip_lib):
    from direct_lidar_odometry_amd import nano_gicp
    return nano_gicp


def _dlo_pair(ng):
    s2s, s2m = ng.NanoGICP(), ng.NanoGICP()
    for e, k, d in ((s2s, 10, 1.0), (s2m, 20, 0.5)):  # odom.cc:100-114 with cfg/params.yaml:54-71
        e.setCorrespondenceRandomness(k); e.setMaxCorrespondenceDistance(d); e.setMaximumIterations(32); e.setTransformationEpsilon(0.01)
    return s2s, s2m


def test_submap_from_device_keyframes_matches_host_route_and_oracle(ng, oracle_mod):
    """odom.cc:1166-1174 (five keyframes, covariances by gicp_s2s with ITS k), :1318-1325 (concatenation in keyframe order),
    :827-834 (hand-over + align): the device route must equal the reference's host route bit for bit and the oracle within
    the north-star tolerance."""
    w = clouds.scan_to_submap(20_000, 5)
    kfs = np.split(w.target, np.cumsum(w.keyframe_sizes)[:-1])
    s2s, s2m = _dlo_pair(ng)
    normals = []
    for kf in kfs:
        s2s.setInputSource(kf); s2s.calculateSourceCovariances()             # odom.cc:1172-1173
        normals.append(s2s.getSourceCovariances())                            # odom.cc:1174 (host route)
        assert s2m.addKeyframe(s2s) == len(normals) - 1                       # device route: nothing leaves the GPU
    assert s2m.numKeyframes() == 5 and [s2m.keyframeSize(i) for i in range(5)] == w.keyframe_sizes
    ids = [0, 1, 2, 3, 4]
    # device route
    s2m.setInputSource(w.source)
    assert s2m.setSubmapKeyframes(ids) is True
    assert s2m.setSubmapKeyframes(ids) is False                                # unchanged keyframe set: no-op (odom.cc:827,1308)
    assert np.array_equal(s2m.targetPoints(), w.target)                       # the concatenation, in the host's point order
    assert np.array_equal(s2m.getTargetCovariances(), np.concatenate(normals))
    s2m.align(w.guess)
    T_dev, it_dev, tr_dev = s2m.getFinalTransformation().copy(), s2m.nr_iterations_, s2m.lm_trace().copy()
    c_dev, _ = s2m.correspondences()
    submap_ms = s2m.stats()["submap_ms"]
    # host route on a fresh engine (odom.cc:830-833)
    ref = _dlo_pair(ng)[1]
    ref.setInputSource(w.source); ref.setInputTarget(w.target); ref.setTargetCovariances(np.concatenate(normals))
    ref.align(w.guess)
    assert np.array_equal(T_dev, ref.getFinalTransformation()) and it_dev == ref.nr_iterations_
    assert np.array_equal(tr_dev[:, [0, 1, 7]], ref.lm_trace()[:, [0, 1, 7]]) and np.allclose(tr_dev, ref.lm_trace(), rtol=1e-9, atol=0)
    assert np.array_equal(c_dev, ref.correspondences()[0])                    # ORIGINAL target indices = offset_k + index in keyframe
    # oracle fed the host-concatenated cloud + covariances
    o = oracle_mod.OracleGICP()
    o.setCorrespondenceRandomness(20); o.setMaxCorrespondenceDistance(0.5); o.setMaximumIterations(32); o.setTransformationEpsilon(0.01)
    o.setInputSource(w.source); o.setInputTarget(w.target); o.setTargetCovariances(np.concatenate(normals))
    o.setSourceCovariances(s2m.getSourceCovariances())
    o.align(w.guess)
    dt, dr = clouds.pose_error(T_dev, o.getFinalTransformation())
    assert dt <= 1e-4 and dr <= 1e-4 and it_dev == o.nr_iterations and s2m.converged_ == o.converged
    print(f"submap of 5 x 20k assembled + indexed on the device in {submap_ms:.3f} ms (host wall)")


def test_submap_subsets_order_and_errors(ng, oracle_mod):
    w = clouds.scan_to_submap(6_000, 4)
    kfs = np.split(w.target, np.cumsum(w.keyframe_sizes)[:-1])
    s2s, s2m = _dlo_pair(ng)
    normals = []
    for kf in kfs:
        s2s.setInputSource(kf); s2s.calculateSourceCovariances(); normals.append(s2s.getSourceCovariances())
        s2m.addKeyframe(s2s)
    s2m.setInputSource(w.source)
    for ids in ([0], [1, 3], [3, 0, 2], [0, 1, 2, 3]):
        assert s2m.setSubmapKeyframes(ids)
        assert np.array_equal(s2m.targetPoints(), np.concatenate([kfs[i] for i in ids]))
        assert np.array_equal(s2m.getTargetCovariances(), np.concatenate([normals[i] for i in ids]))
        s2m.align(w.guess)
        ref = _dlo_pair(ng)[1]
        ref.setInputSource(w.source); ref.setInputTarget(np.concatenate([kfs[i] for i in ids])); ref.setTargetCovariances(np.concatenate([normals[i] for i in ids]))
        ref.align(w.guess)
        assert np.array_equal(s2m.getFinalTransformation(), ref.getFinalTransformation())
    # a host target replaces the submap; the same ids then rebuild it (also when the host cloud has the submap's size and the
    # recycled index object its address)
    s2m.setInputTarget(kfs[0]); s2m.setTargetCovariances(normals[0])
    assert s2m.setSubmapKeyframes([0, 1, 2, 3]) is True
    same_size = np.ascontiguousarray(w.target[::-1])
    s2m.setInputTarget(same_size); s2m.setTargetCovariances(np.concatenate(normals)[::-1])
    assert s2m.setSubmapKeyframes([0, 1, 2, 3]) is True
    assert np.array_equal(s2m.targetPoints(), w.target)
    with pytest.raises(ng.NgicpError):
        s2m.setSubmapKeyframes([0, 7])
    with pytest.raises(ng.NgicpError):
        s2m.setSubmapKeyframes([])
    # covariances computed on demand with the producer's k when the producer has none yet (odom.cc:1173)
    s2s.setInputSource(kfs[1])
    kid = s2m.addKeyframe(s2s)
    s2m.setSubmapKeyframes([kid])
    assert np.array_equal(s2m.getTargetCovariances(), normals[1])
    s2m.clearKeyframes()
    assert s2m.numKeyframes() == 0
    with pytest.raises(ng.NgicpError):
        s2m.setSubmapKeyframes([0])


def test_keyframe_from_transformed_device_scan(ng, oracle_mod):
    """odom.cc:971-974 (transformCurrentScan) + 1166-1174 with the submap voxel filter off: the keyframe is the current scan
    (already on the device as gicp_s2s's source) moved by T and given covariances with gicp_s2s's k - without a host visit."""
    w = clouds.scan_to_scan(10_000)
    T = clouds.make_pose((1.0, -0.5, 0.2), (1.0, -2.0, 30.0)).astype(np.float32)
    s2s, s2m = _dlo_pair(ng)
    s2s.setInputSource(w.source)
    kid = s2m.addKeyframeTransformed(s2s, T)
    s2m.setSubmapKeyframes([kid])
    moved = oracle_mod.transform_cloud(w.source, T)
    assert np.array_equal(s2m.targetPoints(), moved)                          # bit-exact float transform (PCL's SSE2 order)
    s2s.setInputSource(moved); s2s.calculateSourceCovariances()               # the reference's route for the same keyframe
    assert np.array_equal(s2m.getTargetCovariances(), s2s.getSourceCovariances())
    ties = np.zeros(len(moved), bool)
    _, d2 = oracle_mod.OracleTree(moved).knn(moved, 11)
    ties = d2[:, 9] == d2[:, 10]
    assert np.abs(s2m.getTargetCovariances() - oracle_mod.covariances(moved, 10))[~ties].max() < 1e-9


def test_transform_cloud_matches_oracle_restatement(ng, oracle_mod):
    """pcl::transformPointCloud with a float matrix (impl/lsq_registration_impl.hpp:114; odom.cc:484,971-974)."""
    w = clouds.scan_to_scan(10_000)
    rng = np.random.default_rng(3)
    g = ng.NanoGICP(); g.setMaxCorrespondenceDistance(1.0)
    g.setInputSource(clouds.to_xyzi(w.source)); g.setInputTarget(w.target)
    for T in (np.eye(4), clouds.make_pose((0.3, 0.1, 0.02), (0.5, -0.3, 2.0)), clouds.make_pose((-12.5, 40.0, 3.3), (170.0, -80.0, 33.0))):
        T = T.astype(np.float32)
        ref = oracle_mod.transform_cloud(w.source, T)
        assert np.array_equal(g.transformSource(T), ref)                      # device-resident source, original point order
        assert np.array_equal(g.transformCloud(clouds.to_xyzi(w.source), T), ref)  # 32-byte PointXYZI stride in, packed out
        other = oracle_mod.transform_cloud(w.source, T, sse_order=False)      # PCL's scalar fallback order: a rounding or two apart
        assert np.abs(ref - other).max() <= 3 * np.spacing(np.abs(ref).max())
        assert np.abs(ref.astype(np.float64) - (w.source.astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3])).max() < 1e-4
    big = (rng.normal(size=(300_001, 3)) * 30).astype(np.float32)             # not a multiple of the block size
    T = clouds.make_pose((1, 2, 3), (10, 20, 30)).astype(np.float32)
    assert np.array_equal(g.transformCloud(big, T), oracle_mod.transform_cloud(big, T))
    assert g.transformCloud(np.zeros((0, 3), np.float32), T).shape == (0, 3)
    # K5: the aligned output cloud of align() is the same kernel with the final float matrix
    out = g.align(want_aligned=True)
    assert np.array_equal(out, oracle_mod.transform_cloud(w.source, g.getFinalTransformation()))


def test_copy_bandwidth_is_measured(ng):
    g = ng.NanoGICP()
    bw = g.measureCopyBandwidth(1 << 28, 5)
    assert 500.0 < bw < 9000.0, bw  # GB/s, read + write; MI355X nominal 8 TB/s, ~6 TB/s achievable


# ------------------------------------------------------------------ f2: scan preprocessing, f4: map voxel filter
def _raw_scan(n=60_000, seed=7):
    """A scan as the driver delivers it: x y z intensity, with NaN returns and points on the robot itself."""
    rng = np.random.default_rng(seed)
    w = clouds.scan_to_scan(10_000)
    pts = np.concatenate([w.source] * (n // len(w.source) + 1))[:n] + rng.normal(0, 0.02, (n, 3)).astype(np.float32)
    cloud = np.zeros((n, 8), np.float32)  # 32-byte pcl::PointXYZI layout: x y z 1 | intensity 0 0 0
    cloud[:, :3] = pts; cloud[:, 3] = 1.0; cloud[:, 4] = rng.uniform(0, 255, n).astype(np.float32)
    cloud[rng.choice(n, 500, replace=False), rng.integers(0, 3, 500)] = np.nan           # invalid returns
    cloud[rng.choice(n, 300, replace=False), :3] = rng.uniform(-0.9, 0.9, (300, 3)).astype(np.float32)  # hits on the robot (inside the crop box)
    return cloud


@pytest.mark.parametrize("stages", [dict(remove_nan=True), dict(remove_nan=True, crop=1.0), dict(remove_nan=True, crop=1.0, leaf=0.25),
                                    dict(remove_nan=False, leaf=0.5), dict(remove_nan=True, leaf=0.05)])
def test_preprocess_scan_matches_oracle_restatement(ng, oracle_mod, stages):
    """dlo::OdomNode::preprocessPoints (odom.cc:443-465; cfg/params.yaml:26-36: crop 1.0 m, scan voxel 0.25 m) against the oracle's
    restatement of removeNaN / CropBox / VoxelGrid (PCL's source is not under /root/reference: parity unpinned) - bit for bit:
    same survivors in the same order, same voxels in ascending index, centroids (x, y, z AND intensity) summed in input order."""
    cloud = _raw_scan()
    g = ng.NanoGICP()
    got = g.preprocessScan(cloud, remove_nan=stages.get("remove_nan", True), crop_size=stages.get("crop", 0.0), voxel_res=stages.get("leaf", 0.0), intensity_col=4)
    ref = oracle_mod.filter_cloud(cloud, stages.get("remove_nan", True), stages.get("crop", 0.0), stages.get("leaf", 0.0), intensity_col=4)
    assert got.shape == ref.shape and got.shape[0] > 100
    assert np.array_equal(got, ref)
    assert np.isfinite(got[:, :3]).all()
    if stages.get("crop") and not stages.get("leaf"):
        assert not (np.abs(got[:, :3]) <= stages["crop"]).all(axis=1).any()  # nothing left inside the box
    if stages.get("leaf"):  # one output point per occupied voxel of the survivors
        surv = oracle_mod.filter_cloud(cloud, stages.get("remove_nan", True), stages.get("crop", 0.0), 0.0, intensity_col=4)
        surv = surv[np.isfinite(surv[:, :3]).all(axis=1)]
        keys = np.floor(surv[:, :3] * np.float32(1.0 / stages["leaf"])).astype(np.int64)
        assert len(np.unique(keys, axis=0)) == len(got)


def test_preprocessed_scan_becomes_the_source_without_a_round_trip(ng, oracle_mod):
    """preprocessPoints followed by setInputSource(current_scan) (odom.cc:443-465, 519): the filtered cloud, still on the device,
    is indexed directly; aligning it must equal aligning the downloaded copy set the ordinary way."""
    cloud = _raw_scan()
    w = clouds.scan_to_scan(10_000)
    a, b = ng.NanoGICP(), ng.NanoGICP()
    for e in (a, b):
        e.setMaxCorrespondenceDistance(1.0); e.setInputTarget(w.target)
    filtered = a.preprocessScan(cloud, True, 1.0, 0.25, intensity_col=4, set_as_source=True)
    b.setInputSource(np.ascontiguousarray(filtered[:, :3]))
    a.align(); b.align()
    assert np.array_equal(a.getFinalTransformation(), b.getFinalTransformation()) and a.nr_iterations_ == b.nr_iterations_
    assert np.array_equal(a.getSourceCovariances(), b.getSourceCovariances())


def test_map_accumulation_and_voxel_filter(ng, oracle_mod):
    """dlo::MapNode (map.cc:100-131): keyframes appended to the map, the whole map voxel-filtered on every publish."""
    w = clouds.scan_to_submap(6_000, 4)
    kfs = np.split(w.target, np.cumsum(w.keyframe_sizes)[:-1])
    rng = np.random.default_rng(3)
    g = ng.NanoGICP()
    host = np.zeros((0, 4), np.float32)
    for i, kf in enumerate(kfs):
        kfi = np.c_[kf, rng.uniform(0, 100, len(kf)).astype(np.float32)]
        g.mapAdd(kfi, intensity_col=3)
        host = np.concatenate([host, kfi])
        assert g.mapSize() == len(host)
        if i % 2 == 1:  # a publish tick: voxelgrid.filter(*dlo_map) replaces the map
            m = g.mapVoxelFilter(0.3)
            host = oracle_mod.filter_cloud(host, False, 0.0, 0.3, intensity_col=3)
            assert m == len(host)
            assert np.array_equal(g.mapGet(), host)
    g.mapClear()
    assert g.mapSize() == 0 and g.mapGet().shape == (0, 4)


def test_map_filter_between_preprocess_and_set_source_is_refused(ng):
    """The scan filter and the map filter share one workspace: a map publish between preprocessPoints and the hand-over of the
    filtered scan must not leave the engine indexing freed or foreign memory as its source."""
    cloud = _raw_scan(20_000)
    g = ng.NanoGICP()
    g.preprocessScan(cloud, True, 1.0, 0.25, intensity_col=4)
    g.mapAdd(np.ascontiguousarray(cloud[np.isfinite(cloud[:, :3]).all(axis=1)][:, [0, 1, 2, 4]]), intensity_col=3)
    assert g.mapVoxelFilter(0.05) > 0
    with pytest.raises(ng.NgicpError) as ei:
        g._ck(g._L.ngicp_set_source_preprocessed(g._h, 0))
    assert ei.value.code == -3  # NGICP_ERR_STATE (include/ngicp.h)
    # and the ordinary order still works afterwards
    filtered = g.preprocessScan(cloud, True, 1.0, 0.25, intensity_col=4, set_as_source=True)
    assert len(filtered) > 100


def test_filtered_keyframe_on_device_equals_host_route(ng, oracle_mod):
    """DLO's shipped configuration (cfg/params.yaml:33-35: voxelFilter.submap.use = true, res = 0.5): the transformed scan is
    voxel-filtered BEFORE it becomes a keyframe (odom.cc:1160-1163), then indexed and given covariances (odom.cc:1172-1174).
    Device route: addKeyframeTransformedFiltered (nothing visits the host).  Host route, as the reference spells it:
    transformPointCloud -> VoxelGrid -> setInputSource -> calculateSourceCovariances -> getSourceCovariances.  Points and
    covariances must agree bit for bit, and with the oracle's restatements."""
    w = clouds.scan_to_scan(30_000)
    T = clouds.make_pose((0.4, -0.2, 0.05), (0.5, -0.3, 3.0)).astype(np.float32)
    leaf, k = 0.5, 10
    s2s, s2m = ng.NanoGICP(), ng.NanoGICP()
    s2s.setCorrespondenceRandomness(k)
    s2s.setInputSource(w.source)
    kid = s2m.addKeyframeTransformedFiltered(s2s, T, leaf)
    assert kid == 0 and 100 < s2m.keyframeSize(kid) < len(w.source)
    s2m.setSubmapKeyframes([kid])
    dev_pts, dev_covs = s2m.targetPoints(), s2m.getTargetCovariances()
    # host route through the same engine pieces
    scan_t = s2s.transformSource(T)                                              # odom.cc:971-974
    filt = ng.NanoGICP().preprocessScan(np.c_[scan_t, np.zeros(len(scan_t), np.float32)], remove_nan=False, crop_size=0.0, voxel_res=leaf, intensity_col=3)
    host = ng.NanoGICP(); host.setCorrespondenceRandomness(k)
    host.setInputSource(np.ascontiguousarray(filt[:, :3])); host.calculateSourceCovariances()
    assert np.array_equal(dev_pts, filt[:, :3])
    assert np.array_equal(dev_covs, host.getSourceCovariances())
    # and against the oracle's restatements of transformPointCloud / VoxelGrid / calculate_covariances
    ref_t = oracle_mod.transform_cloud(w.source, T)
    ref_f = oracle_mod.filter_cloud(np.c_[ref_t, np.zeros(len(ref_t), np.float32)], False, 0.0, leaf, intensity_col=3)
    assert np.array_equal(dev_pts, ref_f[:, :3])
    ref_c = oracle_mod.covariances(np.ascontiguousarray(ref_f[:, :3]), k, 3, 16)
    _, d2 = oracle_mod.OracleTree(np.ascontiguousarray(ref_f[:, :3])).knn(np.ascontiguousarray(ref_f[:, :3]), k + 1, 16)
    ties = d2[:, k - 1] == d2[:, k]
    assert np.abs(dev_covs - ref_c)[~ties].max() < 1e-9
    # leaf <= 0 falls back to the unfiltered device route
    kid2 = s2m.addKeyframeTransformedFiltered(s2s, T, 0.0)
    assert s2m.keyframeSize(kid2) == len(w.source)
    # the producer's own source slot is untouched
    s2s.setInputTarget(w.target); s2s.setMaxCorrespondenceDistance(1.0); s2s.align()
    ref = ng.NanoGICP(); ref.setCorrespondenceRandomness(k); ref.setMaxCorrespondenceDistance(1.0); ref.setInputSource(w.source); ref.setInputTarget(w.target); ref.align()
    assert np.array_equal(s2s.getFinalTransformation(), ref.getFinalTransformation())


def test_map_voxel_filter_one_million_points(ng, oracle_mod):
    """dlo::MapNode at map scale (map.cc:100-131): ten 100k-point keyframes accumulated, the 1M-point map voxel-filtered twice (two
    publish ticks) - the engine's own stable radix sort over 20-odd key bits, against the oracle's restatement, bit for bit."""
    w = clouds.scan_to_submap(100_000, 10)
    assert len(w.target) == 1_000_000
    rng = np.random.default_rng(11)
    cloud = np.c_[w.target, rng.uniform(0, 255, len(w.target)).astype(np.float32)]
    g = ng.NanoGICP()
    lo = 0
    for n in w.keyframe_sizes:
        g.mapAdd(np.ascontiguousarray(cloud[lo:lo + n]), intensity_col=3)
        lo += n
    assert g.mapSize() == 1_000_000
    host = cloud
    for leaf in (0.3, 0.5):
        m = g.mapVoxelFilter(leaf)
        host = oracle_mod.filter_cloud(host, False, 0.0, leaf, intensity_col=3)
        assert m == len(host) and 1000 < m < 1_000_000
        assert np.array_equal(g.mapGet(), host)
    # a leaf so small that the voxel indices would overflow an int: PCL warns and leaves the cloud as it is
    before = g.mapGet()
    assert g.mapVoxelFilter(1e-6) == len(before)
    assert np.array_equal(g.mapGet(), before)
