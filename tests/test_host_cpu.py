"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/ngicp.h declares,
fails loudly without a GPU, and the multi-rank host logic works over gloo (world_size 2)."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ngicp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ngicp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib):
    from direct_lidar_odometry_amd import nano_gicp
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in include/ngicp.h but not exported"
    assert sorted(nano_gicp.EXPORTS) == declared  # the Python mirror binds exactly the declared surface
    assert b"gfx950" in hip_lib.ngicp_version()


def test_no_silent_cpu_fallback(hip_lib):
    """Without a GPU the product path must raise, never compute on the CPU."""
    import torch
    from direct_lidar_odometry_amd.nano_gicp import NanoGICP, NgicpError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(NgicpError) as e:
        NanoGICP()
    assert e.value.code == -1 and "device" in str(e.value).lower()
    h = ctypes.c_void_p()
    assert hip_lib.ngicp_create(0, ctypes.byref(h)) == -1 and not h.value
    assert hip_lib.ngicp_destroy(None) == 0  # null handle is harmless


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under direct_lidar_odometry_amd/ may reference it."""
    pkg = os.path.join(ROOT, "direct_lidar_odometry_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_shard_bounds_and_packing():
    from direct_lidar_odometry_amd import sharding as sh
    for n in (0, 1, 7, 100_000, 250_001):
        for w in (1, 2, 3, 8):
            spans = [sh.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sorted(sum((sh.partition_items(13, 4, r) for r in range(4)), [])) == list(range(13))
    rng = np.random.default_rng(0)
    H = rng.normal(size=(6, 6)); H = H + H.T
    b = rng.normal(size=6)
    v = sh.pack_sums(H, b, 3.5, 2.5)
    H2, b2, y0, yi = sh.unpack_sums(v)
    assert np.array_equal(H, H2) and np.array_equal(b, b2) and (y0, yi) == (3.5, 2.5)
    assert sorted(sh.tri21(r, c) for r in range(6) for c in range(r, 6)) == list(range(21))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from direct_lidar_odometry_amd import clouds, sharding as sh
        from oracle import oracle as orc
        with np.load(os.path.join(ROOT, "tests", "golden", "ngicp_small.npz")) as z:
            src, tgt, H_ref, b_ref, err_ref = z["source"], z["target"], z["H"], z["b"], float(z["err"])
        # full covariances (they depend on the whole cloud), then this rank's contiguous source block
        full = orc.OracleGICP(); full.setNumThreads(1); full.setInputSource(src); full.setInputTarget(tgt)
        full.calculateSourceCovariances(); full.calculateTargetCovariances()
        cs, ct = full.getSourceCovariances(), full.getTargetCovariances()
        lo, hi = sh.shard_bounds(len(src), world, rank)
        o = orc.OracleGICP(); o.setNumThreads(1); o.setMaxCorrespondenceDistance(1.0)
        o.setInputSource(src[lo:hi]); o.setInputTarget(tgt)
        o.setSourceCovariances(cs[lo:hi]); o.setTargetCovariances(ct)
        H, b, err = o.linearize(np.eye(4))
        v = torch.from_numpy(sh.pack_sums(H, b, err))
        dist.all_reduce(v, op=dist.ReduceOp.SUM)  # the ONE exchange step of the point-sharded path
        H2, b2, y0, _ = sh.unpack_sums(v.numpy())
        ok = (np.abs(H2 - H_ref).max() <= 1e-11 * np.abs(H_ref).max() and np.abs(b2 - b_ref).max() <= 1e-11 * np.abs(b_ref).max()
              and abs(y0 - err_ref) <= 1e-11 * err_ref)
        # independent-alignment path: every rank aligns its own items, results gathered for reporting
        items = sh.partition_items(3, world, rank)
        Ts = []
        for it in items:
            e = orc.OracleGICP(); e.setNumThreads(1); e.setMaxCorrespondenceDistance(1.0)
            e.setInputSource(src[it::3]); e.setInputTarget(tgt)
            Ts.append(e.align())
        allT = sh.gather_results(Ts, dist)
        n_total = sum(len(a) for a in allT)
        q.put((rank, bool(ok), n_total, [a.shape for a in allT]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_reduction_and_gather(oracle_mod):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok, n_total, shapes in res:
        assert ok, f"rank {rank}: all-reduced H/b/err differ from the unsharded linearisation"
        assert n_total == 3 and shapes == [(2, 4, 4), (1, 4, 4)]


# ------------------------------------------------------------------ point-sharded alignment over gloo
class _OracleShardEngine:
    """A CPU stand-in for the HIP engine's sharded protocol (sharded_begin / pass / step / finish, include/ngicp.h), built on the
    oracle's linearize / compute_error hooks: `pass` leaves {H, b, y0 of a linearisation at the trial pose; yi = error of the trial
    pose under the previous correspondences} in the caller's buffer, `step` advances LsqRegistration's LM state machine
    (impl/lsq_registration_impl.hpp:161-208) on the all-reduced sums and reports `done` with the engine's constant lag of two steps.
    It exists so that sharding.sharded_align - the code that runs over RCCL on the GPU node - is executed and checked here."""
    LAG = 2

    def __init__(self, orc, src_block, tgt, cov_src_block, cov_tgt, max_corr, max_iter=64, trans_eps=5e-4, rot_eps=2e-3):
        from oracle import oracle as om
        self.om = om
        self.o = orc.OracleGICP(); self.o.setNumThreads(1); self.o.setMaxCorrespondenceDistance(max_corr)
        self.o.setInputSource(src_block); self.o.setInputTarget(tgt)
        self.o.setSourceCovariances(cov_src_block); self.o.setTargetCovariances(cov_tgt)
        self.max_iter, self.trans_eps, self.rot_eps = max_iter, trans_eps, rot_eps

    @staticmethod
    def _view(ptr):
        return np.ctypeslib.as_array((ctypes.c_double * 32).from_address(ptr))

    def sharded_begin(self, guess):
        self.x0 = np.asarray(np.eye(4) if guess is None else guess, np.float32).astype(np.float64)
        self.xi = self.x0.copy()
        self.have_lin, self.lam, self.nu, self.iter, self.trial = False, -1.0, 2.0, 0, 0
        self.done, self.converged, self.nr_iterations = False, False, 0
        self.flags = []

    def sharded_pass(self, ptr, stream=0):
        from direct_lidar_odometry_amd import sharding as sh
        v = self._view(ptr)
        if self.done:
            return
        yi = self.o.compute_error(self.xi) if self.have_lin else 0.0   # K4: previous correspondences (stale on purpose)
        H, b, y0 = self.o.linearize(self.xi)                           # K2 + K3, speculative: adopted when the trial is accepted
        v[:] = sh.pack_sums(H, b, y0, yi)

    def _trial(self):
        d = self.om.ldlt6_solve(self.H + self.lam * np.eye(6), -self.b)
        self.d = d
        self.delta = np.eye(4); self.delta[:3, :3] = self.om.so3_exp(d[:3]); self.delta[:3, 3] = d[3:]
        self.xi = self.delta @ self.x0

    def _is_converged(self):
        return max((np.abs(self.delta[:3, :3] - np.eye(3)) / self.rot_eps).max(), (np.abs(self.delta[:3, 3]) / self.trans_eps).max()) < 1

    def sharded_step(self, ptr, stream=0) -> bool:
        from direct_lidar_odometry_amd import sharding as sh
        if not self.done:
            H, b, y0, yi = sh.unpack_sums(self._view(ptr))
            if not self.have_lin:
                self.H, self.b, self.y0, self.have_lin = H, b, y0, True
                self.lam = 1e-9 * np.abs(np.diag(H)).max()
                self._trial()
            else:
                rho = (self.y0 - yi) / (self.d @ (self.lam * self.d - self.b))
                if rho < 0:  # rejected: the reference keeps x0 and its linearisation (the oracle's correspondences moved: restore them)
                    if self._is_converged():
                        self.converged = self.done = True
                    else:
                        self.lam *= self.nu; self.nu *= 2; self.trial += 1
                        if self.trial >= 10:
                            self.done = True
                        else:
                            self.o.linearize(self.x0); self._trial()
                else:
                    self.x0 = self.xi
                    self.lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                    self.converged = self._is_converged()
                    self.iter += 1
                    if self.converged or self.iter >= self.max_iter:
                        self.done = True
                    else:
                        self.H, self.b, self.y0 = H, b, y0
                        self.nr_iterations, self.nu, self.trial = self.iter, 2.0, 0
                        self._trial()
        self.flags.append(self.done)
        return self.flags[-1 - self.LAG] if len(self.flags) > self.LAG else False

    def sharded_finish(self):
        return self.x0.astype(np.float32)


def _gloo_sharded_align_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from direct_lidar_odometry_amd import sharding as sh
        from oracle import oracle as orc
        with np.load(os.path.join(ROOT, "tests", "golden", "ngicp_small.npz")) as z:
            src, tgt, guess = z["source"], z["target"], z["guess"]
            T_ref, it_ref = z["final_T"], int(z["nr_iterations"])
        full = orc.OracleGICP(); full.setNumThreads(1); full.setMaxCorrespondenceDistance(1.0); full.setInputSource(src); full.setInputTarget(tgt)
        full.calculateSourceCovariances(); full.calculateTargetCovariances()
        cs, ct = full.getSourceCovariances(), full.getTargetCovariances()
        lo, hi = sh.shard_bounds(len(src), world, rank)
        e = _OracleShardEngine(orc, src[lo:hi], tgt, cs[lo:hi], ct, 1.0)
        T = sh.sharded_align(e, guess, dist, torch.device("cpu"))
        # every rank must end on the identical pose without a broadcast
        t = torch.from_numpy(T.astype(np.float64).copy()); ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        same = all(torch.equal(ts[0], x) for x in ts)
        q.put((rank, T, e.nr_iterations, e.converged, bool(same), len(e.flags), T_ref, it_ref))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_point_sharded_align(oracle_mod):
    """sharding.sharded_align (the driver of the RCCL path) run for real over gloo, 2 ranks: source points split, one 256-byte
    all-reduce per pass, identical LM state machine on both ranks - against the unsharded alignment of the golden fixture."""
    import torch.multiprocessing as mp
    from direct_lidar_odometry_amd import clouds
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_sharded_align_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, T, nit, conv, same, steps, T_ref, it_ref in res:
        assert same, "ranks ended on different poses"
        dt, dr = clouds.pose_error(T, T_ref)
        assert dt <= 1e-6 and dr <= 1e-6, (dt, dr)         # the sharded sums differ from the unsharded ones only in summation order
        assert nit == it_ref and conv
        assert steps == (it_ref + 1) + 1 + _OracleShardEngine.LAG  # one pass per LM trial + the first linearisation + the reporting lag


class _OracleCovShardEngine:
    """Stand-in for the HIP engine's covsShard* protocol on the CPU: the oracle computes the covariances of a block of points
    against the WHOLE cloud (per-point k-NN), the packed [n][6] rows live in a numpy array that torch aliases."""

    def __init__(self, orc, cloud, k):
        self.orc, self.cloud, self.k = orc, np.ascontiguousarray(cloud, np.float32), k
        self.full = orc.covariances(self.cloud, k, 3, 1)  # (what a rank would compute for its block; sliced below)
        self.rows = np.zeros((len(cloud), 6))
        self.committed = False

    def covsShardBegin(self, which):
        return self.rows.ctypes.data, len(self.rows)

    def covsShardTensor(self, which, dev):
        import torch
        return torch.from_numpy(self.rows.reshape(-1))

    def covsShardCompute(self, which, lo, hi, stream=0):
        c = self.full[lo:hi]
        self.rows[lo:hi] = np.stack([c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]], axis=1)

    def covsShardCommit(self, which):
        self.committed = True


def _gloo_sharded_covs_worker(rank, world, port, n, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from direct_lidar_odometry_amd import sharding as sh
        from oracle import oracle as orc
        with np.load(os.path.join(ROOT, "tests", "golden", "ngicp_small.npz")) as z:
            cloud = z["target"][:n]
        e = _OracleCovShardEngine(orc, cloud, 10)
        got_n = sh.sharded_covariances(e, 1, dist, torch.device("cpu"))
        c = e.full
        want = np.stack([c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]], axis=1)
        q.put((rank, got_n, e.committed, bool(np.array_equal(e.rows, want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 1001])  # equal blocks (one all-gather) and ragged blocks (one broadcast per rank)
def test_two_rank_gloo_sharded_covariances(oracle_mod, n):
    """sharding.sharded_covariances (K1 split over ranks) over gloo with 2 ranks: each computes its block of the
    packed covariance array, the blocks are exchanged, both ranks end with the complete, identical set."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_sharded_covs_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, got_n, committed, equal in res:
        assert got_n == n and committed and equal, (rank, got_n, committed, equal)
