"""Independent float64 numpy restatement of the NanoGICP hot path.   *** TEST INFRASTRUCTURE ONLY ***

A second, deliberately different restatement of the reference used to cross-check the C++ oracle
(oracle/ngicp_oracle.cpp) — SURVEY.md §4/§8c "dual restatements".  Where the C++ oracle hand-rolls the
linear algebra, this model uses LAPACK through numpy exactly the way the reference uses Eigen:
  * covariance regularisation via a real SVD  U diag(v) V^T          (impl/nano_gicp_impl.hpp:332-352)
  * Mahalanobis via the literal 4x4 trick RCR(3,3)=1, inverse, (3,3)=0 (impl/nano_gicp_impl.hpp:205-209)
  * H, b via explicit 4x6 Jacobians                                   (impl/nano_gicp_impl.hpp:247-254)
  * 6x6 solve via numpy.linalg.solve                                  (impl/lsq_registration_impl.hpp:172-173)
Neighbour search is brute force with the reference's float32 arithmetic order
(impl/nanoflann_impl.hpp:441-449); only usable for small clouds.
Paths are relative to /root/reference/include/nano_gicp/.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = float(np.finfo(np.float32).max)


def sq_dists_f32(q: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """float32 ((dx*dx + dy*dy) + dz*dz) of one query against all points."""
    d = q.astype(np.float32)[None, :] - pts.astype(np.float32)
    r = d[:, 0] * d[:, 0]
    r = r + d[:, 1] * d[:, 1]
    r = r + d[:, 2] * d[:, 2]
    return r.astype(np.float32)


def knn_bruteforce(queries: np.ndarray, pts: np.ndarray, k: int):
    idx = np.empty((len(queries), k), dtype=np.int32)
    d2 = np.empty((len(queries), k), dtype=np.float32)
    for i, q in enumerate(queries):
        r = sq_dists_f32(q, pts)
        order = np.argsort(r, kind="stable")[:k]
        idx[i] = order
        d2[i] = r[order]
    return idx, d2


def covariances(pts: np.ndarray, k: int = 20, reg: str = "PLANE") -> np.ndarray:
    """impl/nano_gicp_impl.hpp:300-357"""
    n = len(pts)
    idx, _ = knn_bruteforce(pts, pts, k)
    out = np.zeros((n, 4, 4))
    for i in range(n):
        nb = np.ones((4, k))
        nb[:3, :] = pts[idx[i]].astype(np.float64).T
        nb = nb - nb.mean(axis=1, keepdims=True)
        cov = nb @ nb.T / k
        c3 = cov[:3, :3]
        if reg == "NONE":
            out[i] = cov
            continue
        if reg == "FROBENIUS":
            ci = np.linalg.inv(c3 + 1e-3 * np.eye(3))
            out[i, :3, :3] = np.linalg.inv(ci / np.linalg.norm(ci))
            continue
        U, S, Vt = np.linalg.svd(c3)
        if reg == "PLANE":
            vals = np.array([1.0, 1.0, 1e-3])
        elif reg == "MIN_EIG":
            vals = np.maximum(S, 1e-3)
        else:  # NORMALIZED_MIN_EIG
            vals = np.maximum(S / S.max(), 1e-3)
        out[i, :3, :3] = U @ np.diag(vals) @ Vt
    return out


def so3_exp(w: np.ndarray) -> np.ndarray:
    """gicp/so3.hpp:99-118 + Eigen Quaternion::toRotationMatrix (no normalisation)."""
    theta_sq = float(w @ w)
    if theta_sq < 1e-10:
        tq = theta_sq * theta_sq
        imag = 0.5 - theta_sq / 48.0 + tq / 3840.0
        real = 1.0 - theta_sq / 8.0 + tq / 384.0
    else:
        theta = np.sqrt(theta_sq)
        imag = np.sin(0.5 * theta) / theta
        real = np.cos(0.5 * theta)
    qw, qx, qy, qz = real, imag * w[0], imag * w[1], imag * w[2]
    return np.array([
        [1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
        [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
        [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)],
    ])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


class NumpyGICP:
    def __init__(self, source, target, cov_src, cov_tgt, max_corr_dist=FLT_MAX, max_iter=64, trans_eps=5e-4, rot_eps=2e-3,
                 lm_max_iter=10, lm_init_lambda_factor=1e-9):
        self.src = np.asarray(source, dtype=np.float32)
        self.tgt = np.asarray(target, dtype=np.float32)
        self.ca, self.cb = cov_src, cov_tgt
        self.gate = float(max_corr_dist) * float(max_corr_dist)
        self.max_iter, self.trans_eps, self.rot_eps = max_iter, trans_eps, rot_eps
        self.lm_max_iter, self.lm_factor = lm_max_iter, lm_init_lambda_factor
        self.trace = []

    def update_correspondences(self, T):  # impl/nano_gicp_impl.hpp:174-211
        Tf = T.astype(np.float32)
        n = len(self.src)
        self.corr = np.full(n, -1, dtype=np.int64)
        self.sqd = np.zeros(n, dtype=np.float32)
        self.mahal = np.zeros((n, 4, 4))
        for i in range(n):
            p = self.src[i]
            q = np.empty(3, dtype=np.float32)
            for r in range(3):
                q[r] = np.float32(np.float32(np.float32(Tf[r, 0] * p[0]) + np.float32(Tf[r, 1] * p[1])) + np.float32(Tf[r, 2] * p[2])) + Tf[r, 3]
            d = sq_dists_f32(q, self.tgt)
            j = int(np.argmin(d))
            self.sqd[i] = d[j]
            if float(d[j]) < self.gate:
                self.corr[i] = j
                RCR = self.cb[j] + T @ self.ca[i] @ T.T
                RCR[3, 3] = 1.0
                M = np.linalg.inv(RCR)
                M[3, 3] = 0.0
                self.mahal[i] = M

    def accumulate(self, T, want=True):  # impl/nano_gicp_impl.hpp:225-270 / 273-296
        H = np.zeros((6, 6))
        b = np.zeros(6)
        err = 0.0
        for i in range(len(self.src)):
            j = self.corr[i]
            if j < 0:
                continue
            a = np.append(self.src[i].astype(np.float64), 1.0)
            bb = np.append(self.tgt[j].astype(np.float64), 1.0)
            ta = T @ a
            e = bb - ta
            err += e @ self.mahal[i] @ e
            if want:
                J = np.zeros((4, 6))
                J[:3, :3] = skew(ta[:3])
                J[:3, 3:] = -np.eye(3)
                H += J.T @ self.mahal[i] @ J
                b += J.T @ self.mahal[i] @ e
        return H, b, err

    def linearize(self, T):
        self.update_correspondences(T)
        return self.accumulate(T)

    def is_converged(self, delta):  # impl/lsq_registration_impl.hpp:118-127
        R = np.abs(delta[:3, :3] - np.eye(3)) / self.rot_eps
        t = np.abs(delta[:3, 3]) / self.trans_eps
        return max(R.max(), t.max()) < 1

    def align(self, guess):  # impl/lsq_registration_impl.hpp:89-115,161-208
        x0 = np.asarray(guess, dtype=np.float32).astype(np.float64)
        lam = -1.0
        converged = False
        nr = 0
        for it in range(self.max_iter):
            if converged:
                break
            nr = it
            H, b, y0 = self.linearize(x0)
            if lam < 0:
                lam = self.lm_factor * np.abs(np.diag(H)).max()
            nu = 2.0
            ok = False
            delta = np.eye(4)
            for j in range(self.lm_max_iter):
                d = np.linalg.solve(H + lam * np.eye(6), -b)
                delta = np.eye(4)
                delta[:3, :3] = so3_exp(d[:3])
                delta[:3, 3] = d[3:]
                xi = delta @ x0
                _, _, yi = self.accumulate(xi, want=False)
                rho = (y0 - yi) / (d @ (lam * d - b))
                self.trace.append((it, j, y0, yi, rho, lam, np.linalg.norm(d), 0.0 if rho < 0 else 1.0))
                if rho < 0:
                    if self.is_converged(delta):
                        ok = True
                        break
                    lam *= nu
                    nu *= 2
                    continue
                x0 = xi
                lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                ok = True
                break
            if not ok:
                break
            converged = self.is_converged(delta)
        self.converged, self.nr_iterations = converged, nr
        return x0.astype(np.float32)
