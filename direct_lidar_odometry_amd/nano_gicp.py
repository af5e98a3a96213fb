"""Python host-side mirror of nano_gicp::NanoGICP<PointXYZI,PointXYZI> on top of the C ABI.

Same public method names, argument meaning and failure behaviour as the reference class
(/root/reference/include/nano_gicp/nano_gicp.hpp:79-125, lsq_registration.hpp:75-89 and the
pcl::Registration setters DLO uses, /root/reference/src/dlo/odom.cc:100-120), so parity tests read
like tests of the reference.  All compute goes through libngicp_hip.so (hand-written HIP, gfx950);
there is NO CPU fallback: a missing library or GPU raises.

Clouds are numpy float32 arrays of shape (N, C), C >= 3 (C = 8 is the 32-byte pcl::PointXYZI layout);
4x4 matrices are numpy row-major on this side and converted to Eigen's column-major at the boundary;
covariances are (N, 4, 4) float64 like std::vector<Eigen::Matrix4d>.
"""
from __future__ import annotations

import ctypes as C
import enum
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("NGICP_LIB") or os.path.join(_HERE, "libngicp_hip.so")  # NGICP_LIB: another build of the same HIP library (tuning experiments)

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int)

FLT_MAX = float(np.finfo(np.float32).max)


class RegularizationMethod(enum.IntEnum):  # gicp/gicp_settings.hpp:47
    NONE = 0
    MIN_EIG = 1
    NORMALIZED_MIN_EIG = 2
    PLANE = 3
    FROBENIUS = 4


class LSQ_OPTIMIZER_TYPE(enum.IntEnum):  # lsq_registration.hpp:54
    GaussNewton = 0
    LevenbergMarquardt = 1


class NgicpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ngicp error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("align_ms", C.c_double), ("loop_ms", C.c_double), ("pass_ms_total", C.c_double), ("passes", C.c_int),
                ("outer_iterations", C.c_int), ("lm_trials", C.c_int), ("mean_candidates", C.c_double), ("valid_fraction", C.c_double),
                ("index_build_ms", C.c_double), ("covariance_ms", C.c_double), ("upload_ms", C.c_double), ("voxel_size", C.c_double),
                ("grid_dims", C.c_int * 3), ("lanes_per_query", C.c_int), ("passes_timed", C.c_int), ("n_src", C.c_longlong), ("n_tgt", C.c_longlong), ("staged_fraction", C.c_double), ("submap_ms", C.c_double),
                ("device_allocs", C.c_longlong), ("host_wait_spins", C.c_longlong)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["grid_dims"] = list(self.grid_dims)
        return d


# every symbol include/ngicp.h declares (tests check the library exports them all)
EXPORTS = [
    "ngicp_create", "ngicp_destroy", "ngicp_last_error", "ngicp_version", "ngicp_set_params", "ngicp_set_tuning",
    "ngicp_set_source", "ngicp_register_source", "ngicp_set_target", "ngicp_clear_source", "ngicp_clear_target",
    "ngicp_share_source_index", "ngicp_swap_source_target", "ngicp_compute_source_covs", "ngicp_compute_target_covs",
    "ngicp_copy_source_covs", "ngicp_clear_source_covs", "ngicp_clear_target_covs", "ngicp_source_covs_size",
    "ngicp_target_covs_size", "ngicp_get_source_covs", "ngicp_get_target_covs", "ngicp_set_source_covs",
    "ngicp_set_target_covs", "ngicp_align", "ngicp_linearize", "ngicp_compute_error", "ngicp_get_correspondences",
    "ngicp_target_knn", "ngicp_get_lm_trace", "ngicp_get_stats", "ngicp_set_profiling", "ngicp_sharded_begin",
    "ngicp_sharded_pass", "ngicp_sharded_step", "ngicp_sharded_finish",
    "ngicp_keyframe_add", "ngicp_keyframe_add_transformed", "ngicp_keyframe_add_transformed_filtered", "ngicp_keyframe_count", "ngicp_keyframe_size", "ngicp_keyframe_clear",
    "ngicp_submap_set", "ngicp_get_target_points", "ngicp_transform_source", "ngicp_transform_cloud", "ngicp_measure_copy_bandwidth",
    "ngicp_preprocess_scan", "ngicp_set_source_preprocessed", "ngicp_map_add", "ngicp_map_voxel_filter", "ngicp_map_size", "ngicp_map_get",
    "ngicp_map_clear", "ngicp_math_selftest", "ngicp_set_host_wait", "ngicp_covs_shard_begin", "ngicp_covs_shard_compute", "ngicp_covs_shard_commit",
]

_lib = None


def load_library() -> C.CDLL:
    """Load the HIP extension.  Fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; importing torch first makes
    # this extension bind to that same runtime (loading /opt/rocm's copy first leaves torch without GPUs).
    import torch  # noqa: F401  (plumbing: streams, device memory for collectives, torch.distributed)
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} not found: build it with `python -m direct_lidar_odometry_amd.build` "
                          "(or __graft_entry__.build()); this package has no CPU fallback")
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.ngicp_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ngicp_destroy.argtypes = [vp]
    L.ngicp_last_error.argtypes = [vp]
    L.ngicp_last_error.restype = C.c_char_p
    L.ngicp_version.restype = C.c_char_p
    L.ngicp_set_params.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
    L.ngicp_set_tuning.argtypes = [vp, C.c_double, C.c_int]
    L.ngicp_set_host_wait.argtypes = [vp, C.c_int]
    L.ngicp_covs_shard_begin.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.ngicp_covs_shard_compute.argtypes = [vp, C.c_int, C.c_size_t, C.c_size_t, vp]
    L.ngicp_covs_shard_commit.argtypes = [vp, C.c_int]
    for n in ("ngicp_set_source", "ngicp_register_source", "ngicp_set_target"):
        getattr(L, n).argtypes = [vp, c_f32p, C.c_size_t, C.c_size_t, C.c_uint64]
    for n in ("ngicp_clear_source", "ngicp_clear_target", "ngicp_swap_source_target", "ngicp_compute_source_covs",
              "ngicp_compute_target_covs", "ngicp_clear_source_covs", "ngicp_clear_target_covs"):
        getattr(L, n).argtypes = [vp]
    L.ngicp_share_source_index.argtypes = [vp, vp]
    L.ngicp_copy_source_covs.argtypes = [vp, vp]
    L.ngicp_source_covs_size.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.ngicp_target_covs_size.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.ngicp_get_source_covs.argtypes = [vp, c_f64p]
    L.ngicp_get_target_covs.argtypes = [vp, c_f64p]
    L.ngicp_set_source_covs.argtypes = [vp, c_f64p, C.c_size_t]
    L.ngicp_set_target_covs.argtypes = [vp, c_f64p, C.c_size_t]
    L.ngicp_align.argtypes = [vp, c_f32p, c_f32p, c_i32p, c_i32p, c_f64p, c_f32p, C.c_size_t]
    L.ngicp_linearize.argtypes = [vp, c_f64p, c_f64p, c_f64p, c_f64p]
    L.ngicp_compute_error.argtypes = [vp, c_f64p, c_f64p]
    L.ngicp_get_correspondences.argtypes = [vp, c_i32p, c_f32p]
    L.ngicp_target_knn.argtypes = [vp, c_f32p, C.c_size_t, C.c_size_t, C.c_int, c_i32p, c_f32p]
    L.ngicp_get_lm_trace.argtypes = [vp, c_f64p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ngicp_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.ngicp_set_profiling.argtypes = [vp, C.c_int]
    L.ngicp_sharded_begin.argtypes = [vp, c_f32p]
    L.ngicp_sharded_pass.argtypes = [vp, vp, vp]
    L.ngicp_sharded_step.argtypes = [vp, vp, vp, c_i32p]
    L.ngicp_sharded_finish.argtypes = [vp, c_f32p, c_i32p, c_i32p, c_f64p]
    L.ngicp_keyframe_add.argtypes = [vp, vp, c_i32p]
    L.ngicp_keyframe_add_transformed.argtypes = [vp, vp, c_f32p, c_i32p]
    L.ngicp_keyframe_add_transformed_filtered.argtypes = [vp, vp, c_f32p, C.c_float, c_i32p]
    L.ngicp_keyframe_count.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.ngicp_keyframe_size.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.ngicp_keyframe_clear.argtypes = [vp]
    L.ngicp_submap_set.argtypes = [vp, c_i32p, C.c_size_t, c_i32p]
    L.ngicp_get_target_points.argtypes = [vp, c_f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ngicp_transform_source.argtypes = [vp, c_f32p, c_f32p, C.c_size_t]
    L.ngicp_transform_cloud.argtypes = [vp, c_f32p, C.c_size_t, C.c_size_t, c_f32p, c_f32p, C.c_size_t]
    L.ngicp_measure_copy_bandwidth.argtypes = [vp, C.c_size_t, C.c_int, c_f64p]
    L.ngicp_preprocess_scan.argtypes = [vp, c_f32p, C.c_size_t, C.c_size_t, C.c_long, C.c_int, C.c_float, C.c_float, c_f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ngicp_set_source_preprocessed.argtypes = [vp, C.c_uint64]
    L.ngicp_map_add.argtypes = [vp, c_f32p, C.c_size_t, C.c_size_t, C.c_long]
    L.ngicp_map_voxel_filter.argtypes = [vp, C.c_float, C.POINTER(C.c_size_t)]
    L.ngicp_map_size.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.ngicp_map_get.argtypes = [vp, c_f32p, C.c_size_t]
    L.ngicp_map_clear.argtypes = [vp]
    L.ngicp_math_selftest.argtypes = [vp, C.c_int, c_f64p, C.c_size_t, c_f64p]
    _lib = L
    return L


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


def _cloud(a) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] < 3 or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must have shape (N, C>=3)")
    return a


def _colmajor16(T, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=dtype).T.reshape(16))


class NanoGICP:
    """Drop-in mirror of nano_gicp::NanoGICP on one MI355X (see module docstring)."""

    def __init__(self, device: int = 0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.ngicp_create(device, C.byref(h))
        if rc != 0:
            raise NgicpError(rc, (self._L.ngicp_last_error(None) or b"").decode())
        self._h = h
        self.device = device
        # defaults: impl/nano_gicp_impl.hpp:50-64, impl/lsq_registration_impl.hpp:50-63
        self._p = dict(k=20, max_corr_dist=FLT_MAX, max_iter=64, trans_eps=5e-4, rot_eps=2e-3,
                       optimizer=int(LSQ_OPTIMIZER_TYPE.LevenbergMarquardt), lm_max_iter=10, lm_init_lambda_factor=1e-9,
                       regularization=int(RegularizationMethod.PLANE), num_threads=0)
        self._src = None
        self._tgt = None
        self.final_transformation_ = np.eye(4, dtype=np.float32)
        self.converged_ = False
        self.nr_iterations_ = 0
        self.final_hessian_ = np.eye(6)

    def close(self):
        if getattr(self, "_h", None):
            self._L.ngicp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing ----
    def _ck(self, rc: int):
        if rc != 0:
            raise NgicpError(rc, (self._L.ngicp_last_error(self._h) or b"").decode())

    def _push(self):
        p = self._p
        self._ck(self._L.ngicp_set_params(self._h, p["k"], p["max_corr_dist"], p["max_iter"], p["trans_eps"], p["rot_eps"], p["optimizer"],
                                          p["lm_max_iter"], p["lm_init_lambda_factor"], p["regularization"], p["num_threads"]))

    # ---- NanoGICP / LsqRegistration / pcl::Registration setters ----
    def setNumThreads(self, n: int): self._p["num_threads"] = int(n); self._push()                  # impl/nano_gicp_impl.hpp:70-78
    def setCorrespondenceRandomness(self, k: int): self._p["k"] = int(k); self._push()              # :81-83
    def setRegularizationMethod(self, m): self._p["regularization"] = int(m); self._push()          # :86-88
    def setMaxCorrespondenceDistance(self, d: float): self._p["max_corr_dist"] = float(d); self._push()
    def setMaximumIterations(self, n: int): self._p["max_iter"] = int(n); self._push()
    def setTransformationEpsilon(self, e: float): self._p["trans_eps"] = float(e); self._push()
    def setRotationEpsilon(self, e: float): self._p["rot_eps"] = float(e); self._push()             # impl/lsq_registration_impl.hpp:69-71
    def setInitialLambdaFactor(self, f: float): self._p["lm_init_lambda_factor"] = float(f); self._push()  # :74-76
    def setOptimizer(self, t): self._p["optimizer"] = int(t); self._push()
    def setLMMaxIterations(self, n: int): self._p["lm_max_iter"] = int(n); self._push()
    # accepted and ignored, like the reference (odom.cc:104-106,112-120)
    def setEuclideanFitnessEpsilon(self, e): pass
    def setRANSACIterations(self, n): pass
    def setRANSACOutlierRejectionThreshold(self, t): pass
    def setSearchMethodSource(self, tree=None, force_no_recompute=True): pass
    def setSearchMethodTarget(self, tree=None, force_no_recompute=True): pass
    def setDebugPrint(self, on: bool): self._debug = bool(on)

    def setHostWaitMode(self, mode: int):
        """0: the calling thread polls the device without giving its core up (default); 1: it yields between polls."""
        self._ck(self._L.ngicp_set_host_wait(self._h, int(mode)))

    def setTuning(self, voxel_size: float = 0.0, lanes_per_query: int = 0):
        self._ck(self._L.ngicp_set_tuning(self._h, float(voxel_size), int(lanes_per_query)))

    # ---- clouds ----
    def _set(self, fn, cloud, identity):
        c = _cloud(cloud)
        ident = int(identity) if identity is not None else int(c.ctypes.data)
        self._ck(getattr(self._L, fn)(self._h, _p(c, c_f32p), c.shape[0], c.strides[0], ident))
        return c

    def setInputSource(self, cloud, identity=None): self._src = self._set("ngicp_set_source", cloud, identity)         # impl/nano_gicp_impl.hpp:121-129
    def registerInputSource(self, cloud, identity=None): self._src = self._set("ngicp_register_source", cloud, identity)  # :113-118
    def setInputTarget(self, cloud, identity=None): self._tgt = self._set("ngicp_set_target", cloud, identity)         # :132-139

    def clearSource(self):
        self._ck(self._L.ngicp_clear_source(self._h)); self._src = None

    def clearTarget(self):
        self._ck(self._L.ngicp_clear_target(self._h)); self._tgt = None

    def swapSourceAndTarget(self):                                                                                      # :91-98
        self._ck(self._L.ngicp_swap_source_target(self._h))
        self._src, self._tgt = self._tgt, self._src

    def shareSourceIndexFrom(self, other: "NanoGICP"):
        """`this.source_kdtree_ = other.source_kdtree_` (odom.cc:525)."""
        self._ck(self._L.ngicp_share_source_index(self._h, other._h))

    # ---- covariances ----
    def calculateSourceCovariances(self) -> bool:
        self._ck(self._L.ngicp_compute_source_covs(self._h)); return True

    def calculateTargetCovariances(self) -> bool:
        self._ck(self._L.ngicp_compute_target_covs(self._h)); return True

    def copySourceCovariancesFrom(self, other: "NanoGICP"):
        """`this.source_covs_ = other.source_covs_` (odom.cc:815) — stays on the device."""
        self._ck(self._L.ngicp_copy_source_covs(self._h, other._h))

    def clearSourceCovariances(self): self._ck(self._L.ngicp_clear_source_covs(self._h))
    def clearTargetCovariances(self): self._ck(self._L.ngicp_clear_target_covs(self._h))

    def _covs_size(self, fn) -> int:
        n = C.c_size_t(0)
        self._ck(getattr(self._L, fn)(self._h, C.byref(n)))
        return n.value

    def sourceCovariancesSize(self) -> int: return self._covs_size("ngicp_source_covs_size")
    def targetCovariancesSize(self) -> int: return self._covs_size("ngicp_target_covs_size")

    def _get_covs(self, fn, n):
        out = np.empty((n, 16), dtype=np.float64)
        if n:
            self._ck(getattr(self._L, fn)(self._h, _p(out, c_f64p)))
        return out.reshape(n, 4, 4).transpose(0, 2, 1).copy()

    def getSourceCovariances(self): return self._get_covs("ngicp_get_source_covs", self.sourceCovariancesSize())
    def getTargetCovariances(self): return self._get_covs("ngicp_get_target_covs", self.targetCovariancesSize())

    @staticmethod
    def _covs_in(covs):
        c = np.asarray(covs, dtype=np.float64)
        if c.ndim != 3 or c.shape[1:] != (4, 4):
            raise ValueError("covariances must have shape (N,4,4)")
        return np.ascontiguousarray(c.transpose(0, 2, 1).reshape(-1, 16))

    def setSourceCovariances(self, covs):
        c = self._covs_in(covs); self._ck(self._L.ngicp_set_source_covs(self._h, _p(c, c_f64p), c.shape[0]))

    def setTargetCovariances(self, covs):
        c = self._covs_in(covs); self._ck(self._L.ngicp_set_target_covs(self._h, _p(c, c_f64p), c.shape[0]))

    # ---- registration ----
    def align(self, guess=None, want_aligned: bool = False):
        """pcl::Registration::align(output[, guess]).  Returns the aligned cloud (N,3) when asked, else None."""
        g = _colmajor16(np.eye(4) if guess is None else guess, np.float32)
        T = np.empty(16, dtype=np.float32)
        H = np.empty(36, dtype=np.float64)
        conv, nit = C.c_int(0), C.c_int(0)
        aligned, ap, stride = None, None, 0
        if want_aligned:
            if self._src is None:
                raise NgicpError(-3, "no source cloud")
            aligned = np.empty((self._src.shape[0], 3), dtype=np.float32)
            ap, stride = _p(aligned, c_f32p), 12
        rc = self._L.ngicp_align(self._h, _p(g, c_f32p), _p(T, c_f32p), C.byref(conv), C.byref(nit), _p(H, c_f64p), ap, stride)
        self.final_transformation_ = T.reshape(4, 4).T.copy()
        self.converged_ = bool(conv.value)
        self.nr_iterations_ = nit.value
        self.final_hessian_ = H.reshape(6, 6).T.copy()
        self._ck(rc)
        if getattr(self, "_debug", False):
            self._print_lm_table()
        return aligned

    def _print_lm_table(self):
        """setDebugPrint(true): the reference's banner and per-trial table (impl/lsq_registration_impl.hpp:95-99,183-189), printed
        from the engine's trace after the device-resident loop has returned."""
        print("********************************************\n***************** optimize *****************\n********************************************")
        for it, trial, y0, yi, rho, lam, dn, _acc in self.lm_trace():
            if int(trial) == 0:
                print("--- LM optimization ---\n%5s %15s %15s %15s %15s %15s %5s" % ("i", "y0", "yi", "rho", "lambda", "|delta|", "dec"))
            print("%5d %15g %15g %15g %15g %15g %5c" % (int(trial), y0, yi, rho, lam, dn, "x" if rho > 0.0 else " "))

    def getFinalTransformation(self): return self.final_transformation_
    def hasConverged(self) -> bool: return self.converged_
    def getFinalHessian(self): return self.final_hessian_

    # ---- parity hooks ----
    def linearize(self, T):
        t = _colmajor16(T, np.float64)
        H = np.empty(36); b = np.empty(6); e = C.c_double(0)
        self._ck(self._L.ngicp_linearize(self._h, _p(t, c_f64p), _p(H, c_f64p), _p(b, c_f64p), C.byref(e)))
        return H.reshape(6, 6).T.copy(), b, e.value

    def compute_error(self, T) -> float:
        t = _colmajor16(T, np.float64)
        e = C.c_double(0)
        self._ck(self._L.ngicp_compute_error(self._h, _p(t, c_f64p), C.byref(e)))
        return e.value

    def correspondences(self):
        n = self._src.shape[0]
        corr = np.empty(n, dtype=np.int32); sqd = np.empty(n, dtype=np.float32)
        self._ck(self._L.ngicp_get_correspondences(self._h, _p(corr, c_i32p), _p(sqd, c_f32p)))
        return corr, sqd

    def target_knn(self, queries, k: int):
        q = _cloud(queries)
        idx = np.empty((q.shape[0], k), dtype=np.int32); d2 = np.empty((q.shape[0], k), dtype=np.float32)
        self._ck(self._L.ngicp_target_knn(self._h, _p(q, c_f32p), q.shape[0], q.strides[0], k, _p(idx, c_i32p), _p(d2, c_f32p)))
        return idx, d2

    def lm_trace(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._ck(self._L.ngicp_get_lm_trace(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 8))
        if n.value:
            self._ck(self._L.ngicp_get_lm_trace(self._h, _p(out, c_f64p), n.value, C.byref(n)))
        return out

    def setProfiling(self, on):
        """False/0: off; True/1: HIP events around every pass launch; N > 1: around every N-th launch."""
        self._ck(self._L.ngicp_set_profiling(self._h, int(on)))

    def stats(self) -> dict:
        s = Stats()
        self._ck(self._L.ngicp_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    # ---- device-resident keyframes + submap (replaces odom.cc:1174 and :830-833) ----
    def addKeyframe(self, producer: "NanoGICP") -> int:
        kid = C.c_int(-1)
        self._ck(self._L.ngicp_keyframe_add(self._h, producer._h, C.byref(kid)))
        return kid.value

    def addKeyframeTransformed(self, producer: "NanoGICP", T) -> int:
        kid = C.c_int(-1)
        t = _colmajor16(T, np.float32)
        self._ck(self._L.ngicp_keyframe_add_transformed(self._h, producer._h, _p(t, c_f32p), C.byref(kid)))
        return kid.value

    def addKeyframeTransformedFiltered(self, producer: "NanoGICP", T, leaf: float) -> int:
        """DLO's shipped configuration (voxelFilter.submap.use, cfg/params.yaml:33-35): odom.cc:971-974 + 1160-1174 on the device."""
        t = _colmajor16(T, np.float32); kid = C.c_int(-1)
        self._ck(self._L.ngicp_keyframe_add_transformed_filtered(self._h, producer._h, _p(t, c_f32p), float(leaf), C.byref(kid)))
        return kid.value

    def numKeyframes(self) -> int: return self._covs_size("ngicp_keyframe_count")

    def keyframeSize(self, kid: int) -> int:
        n = C.c_size_t(0)
        self._ck(self._L.ngicp_keyframe_size(self._h, int(kid), C.byref(n)))
        return n.value

    def clearKeyframes(self): self._ck(self._L.ngicp_keyframe_clear(self._h))

    def setSubmapKeyframes(self, ids) -> bool:
        """target := concatenation of the keyframes `ids` (cloud + covariances), on the device.  True when rebuilt."""
        a = np.ascontiguousarray(ids, dtype=np.int32)
        changed = C.c_int(0)
        self._ck(self._L.ngicp_submap_set(self._h, _p(a, c_i32p), a.shape[0], C.byref(changed)))
        self._tgt = None
        return bool(changed.value)

    def targetPoints(self) -> np.ndarray:
        """The target cloud as the engine holds it (original point order; for a device submap: the concatenation)."""
        n = C.c_size_t(0)
        self._ck(self._L.ngicp_get_target_points(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), dtype=np.float32)
        if n.value:
            self._ck(self._L.ngicp_get_target_points(self._h, _p(out, c_f32p), 12, C.byref(n)))
        return out

    # ---- rigid transform of clouds (pcl::transformPointCloud with a float matrix) ----
    def transformSource(self, T) -> np.ndarray:
        if self._src is None:
            raise NgicpError(-3, "no source cloud")
        t = _colmajor16(T, np.float32)
        out = np.empty((self._src.shape[0], 3), dtype=np.float32)
        self._ck(self._L.ngicp_transform_source(self._h, _p(t, c_f32p), _p(out, c_f32p), 12))
        return out

    def transformCloud(self, cloud, T) -> np.ndarray:
        c = _cloud(cloud)
        t = _colmajor16(T, np.float32)
        out = np.empty((c.shape[0], 3), dtype=np.float32)
        self._ck(self._L.ngicp_transform_cloud(self._h, _p(c, c_f32p), c.shape[0], c.strides[0], _p(t, c_f32p), _p(out, c_f32p), 12))
        return out

    # ---- scan preprocessing (src/dlo/odom.cc:443-465) and map voxel filter (§8f-4; src/dlo/map.cc:100-131) ----
    @staticmethod
    def _intensity_offset(c: np.ndarray, intensity_col) -> int:
        return -1 if intensity_col is None or intensity_col < 0 else int(intensity_col) * 4

    def preprocessScan(self, cloud, remove_nan: bool = True, crop_size: float = 0.0, voxel_res: float = 0.0, intensity_col=None,
                       set_as_source: bool = False) -> np.ndarray:
        """removeNaN -> CropBox(negative, +-crop_size) -> VoxelGrid(voxel_res); returns (M, 4) {x, y, z, intensity}.  With
        set_as_source the filtered cloud (still on the device) also becomes the source: setInputSource without an upload."""
        c = np.asarray(cloud, dtype=np.float32)
        if c.ndim != 2 or c.shape[1] < 3 or not c.flags.c_contiguous:
            c = np.ascontiguousarray(c, dtype=np.float32)
        out = np.empty((c.shape[0], 4), dtype=np.float32)
        m = C.c_size_t(0)
        self._ck(self._L.ngicp_preprocess_scan(self._h, _p(c, c_f32p), c.shape[0], c.strides[0], self._intensity_offset(c, intensity_col),
                                               1 if remove_nan else 0, float(crop_size), float(voxel_res), _p(out, c_f32p), out.shape[0], C.byref(m)))
        out = out[:m.value].copy()
        if set_as_source:
            self._ck(self._L.ngicp_set_source_preprocessed(self._h, 0))
            self._src = np.ascontiguousarray(out[:, :3])
        return out

    def mapAdd(self, cloud, intensity_col=None):
        c = np.asarray(cloud, dtype=np.float32)
        if c.ndim != 2 or c.shape[1] < 3 or not c.flags.c_contiguous:
            c = np.ascontiguousarray(c, dtype=np.float32)
        self._ck(self._L.ngicp_map_add(self._h, _p(c, c_f32p), c.shape[0], c.strides[0], self._intensity_offset(c, intensity_col)))

    def mapVoxelFilter(self, leaf: float) -> int:
        m = C.c_size_t(0)
        self._ck(self._L.ngicp_map_voxel_filter(self._h, float(leaf), C.byref(m)))
        return m.value

    def mapSize(self) -> int: return self._covs_size("ngicp_map_size")

    def mapGet(self) -> np.ndarray:
        n = self.mapSize()
        out = np.empty((n, 4), dtype=np.float32)
        if n:
            self._ck(self._L.ngicp_map_get(self._h, _p(out, c_f32p), n))
        return out

    def mapClear(self): self._ck(self._L.ngicp_map_clear(self._h))

    def mathSelftest(self, which: int, problems) -> np.ndarray:
        """ngicp_math.h on the device (0 so3_exp, 1 ldlt6_solve, 2 eig3_sym, 3 inv3_sym), one problem per row."""
        a = np.ascontiguousarray(problems, dtype=np.float64)
        out = np.empty((a.shape[0], (9, 6, 12, 6)[which]))
        self._ck(self._L.ngicp_math_selftest(self._h, int(which), _p(a, c_f64p), a.shape[0], _p(out, c_f64p)))
        return out

    def measureCopyBandwidth(self, nbytes: int = 1 << 30, reps: int = 10) -> float:
        """Device float4 stream copy, (read + write) GB/s."""
        v = C.c_double(0)
        self._ck(self._L.ngicp_measure_copy_bandwidth(self._h, int(nbytes), int(reps), C.byref(v)))
        return v.value

    # ---- covariances sharded over ranks (K1 with an all-gather of the packed covariances) ----
    def covsShardBegin(self, which: int):
        """-> (device pointer of the packed [n][6] FP64 set, n).  which: 0 source, 1 target."""
        ptr = C.c_void_p(0); n = C.c_size_t(0)
        self._ck(self._L.ngicp_covs_shard_begin(self._h, int(which), C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def covsShardCompute(self, which: int, lo: int, hi: int, stream: int = 0):
        self._ck(self._L.ngicp_covs_shard_compute(self._h, int(which), int(lo), int(hi), C.c_void_p(stream) if stream else None))

    def covsShardCommit(self, which: int):
        self._ck(self._L.ngicp_covs_shard_commit(self._h, int(which)))

    # ---- point-sharded stepping; buffers are raw device pointers ----
    def sharded_begin(self, guess=None):
        g = _colmajor16(np.eye(4) if guess is None else guess, np.float32)
        self._ck(self._L.ngicp_sharded_begin(self._h, _p(g, c_f32p)))

    def sharded_pass(self, sums_dev_ptr: int, stream: int = 0):
        self._ck(self._L.ngicp_sharded_pass(self._h, C.c_void_p(sums_dev_ptr), C.c_void_p(stream) if stream else None))

    def sharded_step(self, sums_dev_ptr: int, stream: int = 0) -> bool:
        done = C.c_int(0)
        self._ck(self._L.ngicp_sharded_step(self._h, C.c_void_p(sums_dev_ptr), C.c_void_p(stream) if stream else None, C.byref(done)))
        return bool(done.value)

    def sharded_finish(self):
        T = np.empty(16, dtype=np.float32); H = np.empty(36); conv, nit = C.c_int(0), C.c_int(0)
        self._ck(self._L.ngicp_sharded_finish(self._h, _p(T, c_f32p), C.byref(conv), C.byref(nit), _p(H, c_f64p)))
        self.final_transformation_ = T.reshape(4, 4).T.copy()
        self.converged_ = bool(conv.value); self.nr_iterations_ = nit.value
        self.final_hessian_ = H.reshape(6, 6).T.copy()
        return self.final_transformation_


def keyframe_covariances(target, keyframe_sizes, k: int = 20, device: int = 0) -> np.ndarray:
    """Per-keyframe covariances of a submap that is a concatenation of keyframes, concatenated in the same order — DLO's
    `keyframe_normals` / `submap_normals` (src/dlo/odom.cc:1172-1174,1318-1325): each keyframe's covariances come from ITS OWN
    points only.  Used by tests and bench.py to prepare the scan-to-submap workloads."""
    e = NanoGICP(device=device)
    e.setCorrespondenceRandomness(k)
    out, lo = [], 0
    for n in keyframe_sizes:
        e.setInputSource(np.ascontiguousarray(target[lo:lo + n]))
        e.calculateSourceCovariances()
        out.append(e.getSourceCovariances())
        lo += n
    e.close()
    return np.concatenate(out)
