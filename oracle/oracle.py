"""ctypes front-end of the CPU oracle.   *** TEST INFRASTRUCTURE ONLY ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see ngicp_oracle.cpp header).  `OracleGICP` mirrors the public
method names of nano_gicp::NanoGICP (/root/reference/include/nano_gicp/nano_gicp.hpp:79-125)
so parity tests can drive it and the HIP engine with the same call sequence.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_ngicp.so")
_REF = os.path.join(_HERE, "_ref", "libref_nanoflann.so")

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int)


def build(ref: bool = True) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL)
    if ref and os.path.isdir("/root/reference/include"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


def _fp(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build(ref=False)
        L = C.CDLL(_LIB)
        L.orc_tree_build.restype = C.c_void_p
        L.orc_tree_build.argtypes = [c_f32p, C.c_size_t, C.c_size_t]
        L.orc_tree_free.argtypes = [C.c_void_p]
        L.orc_tree_knn.argtypes = [C.c_void_p, c_f32p, C.c_size_t, C.c_size_t, C.c_int, c_i32p, c_f32p, C.c_int]
        L.orc_covariances.argtypes = [c_f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, c_f64p, C.c_int]
        L.orc_so3_exp.argtypes = [c_f64p, c_f64p]
        L.orc_ldlt6_solve.argtypes = [c_f64p, c_f64p, c_f64p]
        L.orc_eig3_sym.argtypes = [c_f64p, c_f64p, c_f64p]
        L.orc_gicp_create.restype = C.c_void_p
        L.orc_gicp_destroy.argtypes = [C.c_void_p]
        L.orc_gicp_set_params.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_gicp_num_threads.argtypes = [C.c_void_p]
        for name in ("set_source", "register_source", "set_target"):
            getattr(L, "orc_gicp_" + name).argtypes = [C.c_void_p, c_f32p, C.c_size_t, C.c_size_t, C.c_uint64]
        L.orc_gicp_share_source_index.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_gicp_copy_source_covs.argtypes = [C.c_void_p, C.c_void_p]
        for name in ("clear_source_covs", "compute_source_covs", "compute_target_covs", "swap_source_target"):
            getattr(L, "orc_gicp_" + name).argtypes = [C.c_void_p]
        for name in ("source_covs_size", "target_covs_size", "trace_rows"):
            f = getattr(L, "orc_gicp_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_size_t
        L.orc_gicp_get_source_covs.argtypes = [C.c_void_p, c_f64p]
        L.orc_gicp_get_target_covs.argtypes = [C.c_void_p, c_f64p]
        L.orc_gicp_set_source_covs.argtypes = [C.c_void_p, c_f64p, C.c_size_t]
        L.orc_gicp_set_target_covs.argtypes = [C.c_void_p, c_f64p, C.c_size_t]
        L.orc_gicp_align.argtypes = [C.c_void_p, c_f32p, c_f32p, c_i32p, c_i32p, c_f64p, c_f32p, C.c_size_t]
        L.orc_gicp_linearize.argtypes = [C.c_void_p, c_f64p, c_f64p, c_f64p, c_f64p]
        L.orc_gicp_compute_error.argtypes = [C.c_void_p, c_f64p, c_f64p]
        L.orc_gicp_get_correspondences.argtypes = [C.c_void_p, c_i32p, c_f32p]
        L.orc_gicp_get_mahalanobis.argtypes = [C.c_void_p, c_f64p]
        L.orc_gicp_get_trace.argtypes = [C.c_void_p, c_f64p]
        L.orc_gicp_lambda.argtypes = [C.c_void_p]
        L.orc_gicp_lambda.restype = C.c_double
        L.orc_gicp_set_debug.argtypes = [C.c_void_p, C.c_int]
        L.orc_transform_cloud.argtypes = [c_f32p, C.c_size_t, C.c_size_t, c_f32p, C.c_int, c_f32p]
        L.orc_filter_cloud.argtypes = [c_f32p, C.c_size_t, C.c_size_t, C.c_long, C.c_int, C.c_float, C.c_float, c_f32p]
        L.orc_filter_cloud.restype = C.c_size_t
        _lib = L
    return _lib


_ref = None


def ref_available() -> bool:
    return os.path.exists(_REF)


def ref_lib() -> C.CDLL:
    global _ref
    if _ref is None:
        L = C.CDLL(_REF)
        L.ref_tree_build.restype = C.c_void_p
        L.ref_tree_build.argtypes = [c_f32p, C.c_size_t, C.c_size_t]
        L.ref_tree_free.argtypes = [C.c_void_p]
        L.ref_tree_knn.argtypes = [C.c_void_p, c_f32p, C.c_size_t, C.c_size_t, C.c_int, c_i32p, c_f32p, C.c_int]
        _ref = L
    return _ref


def _xyz(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a


class _Tree:
    def __init__(self, L, prefix, pts):
        self._L, self._p = L, prefix
        pts = _xyz(pts)
        self._h = getattr(L, prefix + "_tree_build")(_fp(pts, c_f32p), pts.shape[0], pts.shape[1])

    def knn(self, queries, k: int, threads: int = 8):
        q = _xyz(queries)
        idx = np.empty((q.shape[0], k), dtype=np.int32)
        d2 = np.empty((q.shape[0], k), dtype=np.float32)
        rc = getattr(self._L, self._p + "_tree_knn")(self._h, _fp(q, c_f32p), q.shape[0], q.shape[1], k, _fp(idx, c_i32p), _fp(d2, c_f32p), threads)
        assert rc == 0
        return idx, d2

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(self._L, self._p + "_tree_free")(self._h)
            self._h = None


def OracleTree(pts):
    """Own kd-tree restatement."""
    return _Tree(lib(), "orc", pts)


def RefTree(pts):
    """The real reference nanoflann (only in the authoring container / when prebuilt)."""
    return _Tree(ref_lib(), "ref", pts)


def covariances(pts, k: int = 20, reg: int = 3, threads: int = 8) -> np.ndarray:
    pts = _xyz(pts)
    out = np.empty((pts.shape[0], 16), dtype=np.float64)
    rc = lib().orc_covariances(_fp(pts, c_f32p), pts.shape[0], pts.shape[1], k, reg, _fp(out, c_f64p), threads)
    if rc:
        raise RuntimeError(f"orc_covariances rc={rc}")
    return out.reshape(-1, 4, 4).transpose(0, 2, 1)  # column-major -> [i][r][c]


def transform_cloud(pts, T, sse_order: bool = True) -> np.ndarray:
    """pcl::transformPointCloud(in, out, Eigen::Matrix4f) restated (ngicp_oracle.cpp transform_point_pcl; parity unpinned)."""
    pts = _xyz(pts)
    t = np.ascontiguousarray(np.asarray(T, dtype=np.float32).T.reshape(16))
    out = np.empty((pts.shape[0], 3), dtype=np.float32)
    lib().orc_transform_cloud(_fp(pts, c_f32p), pts.shape[0], pts.shape[1], _fp(t, c_f32p), 1 if sse_order else 0, _fp(out, c_f32p))
    return out


def filter_cloud(pts, remove_nan: bool = True, crop_half: float = 0.0, leaf: float = 0.0, intensity_col: int = -1) -> np.ndarray:
    """removeNaN + CropBox(negative) + VoxelGrid restated (ngicp_oracle.cpp "filters"; parity unpinned): (M, 4) {x, y, z, intensity}."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty((pts.shape[0], 4), dtype=np.float32)
    m = lib().orc_filter_cloud(_fp(pts, c_f32p), pts.shape[0], pts.shape[1], intensity_col, 1 if remove_nan else 0, crop_half, leaf, _fp(out, c_f32p))
    return out[:m].copy()


def so3_exp(w) -> np.ndarray:
    w = np.ascontiguousarray(w, dtype=np.float64)
    R = np.empty(9, dtype=np.float64)
    lib().orc_so3_exp(_fp(w, c_f64p), _fp(R, c_f64p))
    return R.reshape(3, 3)


def ldlt6_solve(A, rhs) -> np.ndarray:
    A = np.ascontiguousarray(A, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    x = np.empty(6, dtype=np.float64)
    lib().orc_ldlt6_solve(_fp(A, c_f64p), _fp(rhs, c_f64p), _fp(x, c_f64p))
    return x


def eig3_sym(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    w = np.empty(3)
    V = np.empty(9)
    lib().orc_eig3_sym(_fp(A, c_f64p), _fp(w, c_f64p), _fp(V, c_f64p))
    return w, V.reshape(3, 3)


def _colmajor16(T, dtype):
    return np.ascontiguousarray(np.asarray(T, dtype=dtype).T.reshape(16))


def _covs_in(covs) -> np.ndarray:
    c = np.asarray(covs, dtype=np.float64)
    assert c.ndim == 3 and c.shape[1:] == (4, 4)
    return np.ascontiguousarray(c.transpose(0, 2, 1).reshape(-1, 16))


class OracleGICP:
    """Method-for-method mirror of nano_gicp::NanoGICP for the hot path (see module docstring)."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.orc_gicp_create()
        self.p = dict(k=20, max_corr_dist=float(np.finfo(np.float32).max), max_iter=64, trans_eps=5e-4, rot_eps=2e-3,
                      optimizer=1, lm_max_iter=10, lm_init_lambda_factor=1e-9, regularization=3, num_threads=0)
        self._keep = {}
        self._push()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_gicp_destroy(self.h)
            self.h = None

    def _push(self):
        p = self.p
        self.L.orc_gicp_set_params(self.h, p["k"], p["max_corr_dist"], p["max_iter"], p["trans_eps"], p["rot_eps"], p["optimizer"],
                                   p["lm_max_iter"], p["lm_init_lambda_factor"], p["regularization"], p["num_threads"])

    # --- setters (reference: impl/nano_gicp_impl.hpp:70-88, impl/lsq_registration_impl.hpp:69-81, PCL base) ---
    def setNumThreads(self, n): self.p["num_threads"] = n; self._push()
    def setCorrespondenceRandomness(self, k): self.p["k"] = k; self._push()
    def setRegularizationMethod(self, m): self.p["regularization"] = int(m); self._push()
    def setMaxCorrespondenceDistance(self, d): self.p["max_corr_dist"] = float(d); self._push()
    def setMaximumIterations(self, n): self.p["max_iter"] = int(n); self._push()
    def setTransformationEpsilon(self, e): self.p["trans_eps"] = float(e); self._push()
    def setRotationEpsilon(self, e): self.p["rot_eps"] = float(e); self._push()
    def setInitialLambdaFactor(self, f): self.p["lm_init_lambda_factor"] = float(f); self._push()
    def setOptimizer(self, lm: bool): self.p["optimizer"] = 1 if lm else 0; self._push()
    def setLMMaxIterations(self, n): self.p["lm_max_iter"] = int(n); self._push()
    def setDebugPrint(self, on): self.L.orc_gicp_set_debug(self.h, 1 if on else 0)
    def numThreads(self): return self.L.orc_gicp_num_threads(self.h)

    def _cloud(self, fn, cloud, identity):
        c = _xyz(cloud)
        ident = int(identity) if identity is not None else int(c.ctypes.data)
        rc = getattr(self.L, fn)(self.h, _fp(c, c_f32p), c.shape[0], c.shape[1], ident)
        assert rc == 0
        return c

    def setInputSource(self, cloud, identity=None): self._keep["src"] = self._cloud("orc_gicp_set_source", cloud, identity)
    def registerInputSource(self, cloud, identity=None): self._keep["src"] = self._cloud("orc_gicp_register_source", cloud, identity)
    def setInputTarget(self, cloud, identity=None): self._keep["tgt"] = self._cloud("orc_gicp_set_target", cloud, identity)
    def shareSourceIndexFrom(self, other): self.L.orc_gicp_share_source_index(self.h, other.h)
    def copySourceCovariancesFrom(self, other): self.L.orc_gicp_copy_source_covs(self.h, other.h)
    def clearSourceCovariances(self): self.L.orc_gicp_clear_source_covs(self.h)
    def swapSourceAndTarget(self): self.L.orc_gicp_swap_source_target(self.h)

    def calculateSourceCovariances(self):
        rc = self.L.orc_gicp_compute_source_covs(self.h)
        if rc: raise RuntimeError(f"rc={rc}")
        return True

    def calculateTargetCovariances(self):
        rc = self.L.orc_gicp_compute_target_covs(self.h)
        if rc: raise RuntimeError(f"rc={rc}")
        return True

    def getSourceCovariances(self):
        n = self.L.orc_gicp_source_covs_size(self.h)
        out = np.empty((n, 16))
        self.L.orc_gicp_get_source_covs(self.h, _fp(out, c_f64p))
        return out.reshape(-1, 4, 4).transpose(0, 2, 1).copy()

    def getTargetCovariances(self):
        n = self.L.orc_gicp_target_covs_size(self.h)
        out = np.empty((n, 16))
        self.L.orc_gicp_get_target_covs(self.h, _fp(out, c_f64p))
        return out.reshape(-1, 4, 4).transpose(0, 2, 1).copy()

    def setSourceCovariances(self, covs):
        c = _covs_in(covs)
        self.L.orc_gicp_set_source_covs(self.h, _fp(c, c_f64p), c.shape[0])

    def setTargetCovariances(self, covs):
        c = _covs_in(covs)
        self.L.orc_gicp_set_target_covs(self.h, _fp(c, c_f64p), c.shape[0])

    def align(self, guess=None, want_aligned=False):
        g = _colmajor16(np.eye(4) if guess is None else guess, np.float32)
        T = np.empty(16, dtype=np.float32)
        conv, nit = C.c_int(0), C.c_int(0)
        H = np.empty(36)
        aligned = None
        ap, stride = None, 0
        if want_aligned:
            aligned = np.zeros((self._keep["src"].shape[0], 3), dtype=np.float32)
            ap, stride = _fp(aligned, c_f32p), 3
        rc = self.L.orc_gicp_align(self.h, _fp(g, c_f32p), _fp(T, c_f32p), C.byref(conv), C.byref(nit), _fp(H, c_f64p), ap, stride)
        if rc:
            raise RuntimeError(f"orc_gicp_align rc={rc}")
        self.final_transformation = T.reshape(4, 4).T.copy()
        self.converged = bool(conv.value)
        self.nr_iterations = nit.value
        self.final_hessian = H.reshape(6, 6).T.copy()
        self.aligned = aligned
        return self.final_transformation

    def getFinalTransformation(self): return self.final_transformation
    def hasConverged(self): return self.converged
    def getFinalHessian(self): return self.final_hessian

    def linearize(self, T):
        t = _colmajor16(T, np.float64)
        H = np.empty(36); b = np.empty(6); e = C.c_double(0)
        rc = self.L.orc_gicp_linearize(self.h, _fp(t, c_f64p), _fp(H, c_f64p), _fp(b, c_f64p), C.byref(e))
        if rc: raise RuntimeError(f"rc={rc}")
        return H.reshape(6, 6).T.copy(), b, e.value

    def compute_error(self, T):
        t = _colmajor16(T, np.float64)
        e = C.c_double(0)
        rc = self.L.orc_gicp_compute_error(self.h, _fp(t, c_f64p), C.byref(e))
        if rc: raise RuntimeError(f"rc={rc}")
        return e.value

    def correspondences(self):
        n = self._keep["src"].shape[0]
        corr = np.empty(n, dtype=np.int32); sqd = np.empty(n, dtype=np.float32)
        self.L.orc_gicp_get_correspondences(self.h, _fp(corr, c_i32p), _fp(sqd, c_f32p))
        return corr, sqd

    def mahalanobis(self):
        n = self._keep["src"].shape[0]
        out = np.empty((n, 16))
        self.L.orc_gicp_get_mahalanobis(self.h, _fp(out, c_f64p))
        return out.reshape(-1, 4, 4).transpose(0, 2, 1).copy()

    def lm_trace(self):
        n = self.L.orc_gicp_trace_rows(self.h)
        out = np.empty((n, 8))
        if n: self.L.orc_gicp_get_trace(self.h, _fp(out, c_f64p))
        return out
