// Small fixed-size FP64 linear algebra shared by the HIP kernels and the host driver.
// These stand in for the Eigen calls the reference makes on its hot path
// (/root/reference/include/nano_gicp/impl/nano_gicp_impl.hpp:205-209,320-352,
//  impl/lsq_registration_impl.hpp:147-152,172-179, gicp/so3.hpp:99-118).
// Everything is written with compile-time indices so that device code stays in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define NG_HD __host__ __device__ __forceinline__

namespace ngk {

// Rigid transform, R row-major.  (Eigen::Isometry3d restated.)
struct Pose {
  double R[9];
  double t[3];
};

NG_HD void pose_identity(Pose& p) {
  for (int i = 0; i < 9; ++i) p.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  p.t[0] = p.t[1] = p.t[2] = 0.0;
}

// c = a * b  (delta * x0, impl/lsq_registration_impl.hpp:154,179)
NG_HD void pose_mul(const Pose& a, const Pose& b, Pose& c) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) c.R[i * 3 + j] = a.R[i * 3 + 0] * b.R[0 * 3 + j] + a.R[i * 3 + 1] * b.R[1 * 3 + j] + a.R[i * 3 + 2] * b.R[2 * 3 + j];
    c.t[i] = a.R[i * 3 + 0] * b.t[0] + a.R[i * 3 + 1] * b.t[1] + a.R[i * 3 + 2] * b.t[2] + a.t[i];
  }
}

// gicp/so3.hpp:99-118 followed by Eigen's Quaternion::toRotationMatrix() (no normalisation, as upstream)
NG_HD void so3_exp_matrix(const double w[3], double R[9]) {
  const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = sqrt(theta_sq);
    const double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  const double qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}

// Symmetric 3x3 stored as {xx, xy, xz, yy, yz, zz}.
// Inverse by cofactors: the 3x3 block of the reference's 4x4 inverse (impl/nano_gicp_impl.hpp:205-209).
NG_HD void inv3_sym(const double a[6], double o[6]) {
  const double xx = a[0], xy = a[1], xz = a[2], yy = a[3], yz = a[4], zz = a[5];
  const double c00 = yy * zz - yz * yz;
  const double c01 = xz * yz - xy * zz;
  const double c02 = xy * yz - xz * yy;
  const double det = xx * c00 + xy * c01 + xz * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id;
  o[1] = c01 * id;
  o[2] = c02 * id;
  o[3] = (xx * zz - xz * xz) * id;
  o[4] = (xy * xz - xx * yz) * id;
  o[5] = (xx * yy - xy * xy) * id;
}

// out = sym( R * A * R^T ), A symmetric-6, R row-major 3x3
NG_HD void rotate_sym(const double R[9], const double A[6], double o[6]) {
  // RA = R * A
  const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
  double RA[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double r0 = R[r * 3 + 0], r1 = R[r * 3 + 1], r2 = R[r * 3 + 2];
    RA[r * 3 + 0] = r0 * a00 + r1 * a01 + r2 * a02;
    RA[r * 3 + 1] = r0 * a01 + r1 * a11 + r2 * a12;
    RA[r * 3 + 2] = r0 * a02 + r1 * a12 + r2 * a22;
  }
  o[0] = RA[0] * R[0] + RA[1] * R[1] + RA[2] * R[2];
  o[1] = RA[0] * R[3] + RA[1] * R[4] + RA[2] * R[5];
  o[2] = RA[0] * R[6] + RA[1] * R[7] + RA[2] * R[8];
  o[3] = RA[3] * R[3] + RA[4] * R[4] + RA[5] * R[5];
  o[4] = RA[3] * R[6] + RA[4] * R[7] + RA[5] * R[8];
  o[5] = RA[6] * R[6] + RA[7] * R[7] + RA[8] * R[8];
}

// One Jacobi rotation annihilating a[p][q] of a symmetric 3x3 held in named scalars.
// app, aqq: diagonal entries; apq: the pivot; arp, arq: the two entries coupling the third index r.
// vXp, vXq: eigenvector columns p and q.
NG_HD void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq, double& v0p, double& v0q, double& v1p, double& v1q, double& v2p,
                      double& v2q) {
  if (apq == 0.0) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  app = app - t * apq;
  aqq = aqq + t * apq;
  apq = 0.0;
  const double rp = arp, rq = arq;
  arp = c * rp - s * rq;
  arq = s * rp + c * rq;
  double a, b;
  a = v0p; b = v0q; v0p = c * a - s * b; v0q = s * a + c * b;
  a = v1p; b = v1q; v1p = c * a - s * b; v1q = s * a + c * b;
  a = v2p; b = v2q; v2p = c * a - s * b; v2q = s * a + c * b;
}

// Cyclic Jacobi eigen-decomposition of symmetric 3x3 {xx,xy,xz,yy,yz,zz}.
// Stands in for JacobiSVD<Matrix3d> (impl/nano_gicp_impl.hpp:332): for symmetric PSD input the
// singular vectors are the eigenvectors and the singular values the |eigenvalues|.
// Output: w[3] eigenvalues (unsorted), V row-major with eigenvectors as COLUMNS.
NG_HD void eig3_sym(const double A[6], double w[3], double V[9]) {
  double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 24; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    const double dg = fabs(a00) + fabs(a11) + fabs(a22);
    if (off <= 1e-300 || off <= 1e-22 * dg) break;
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);  // (p,q)=(0,1), r=2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);  // (0,2), r=1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);  // (1,2), r=0
  }
  w[0] = a00; w[1] = a11; w[2] = a22;
  V[0] = v00; V[1] = v01; V[2] = v02;
  V[3] = v10; V[4] = v11; V[5] = v12;
  V[6] = v20; V[7] = v21; V[8] = v22;
}

// 6x6 symmetric solve by LDL^T with diagonal pivoting (Eigen::LDLT<Matrix6d>::solve stand-in,
// impl/lsq_registration_impl.hpp:147-148,172-173).  A is row-major and is destroyed.
// Zero pivots contribute zero to the solution, like Eigen's solve.  Every array index is a
// compile-time constant after unrolling (runtime pivots are applied through predicated swaps), so on
// the device the whole factorisation lives in registers.
#define NG_SWAP(a, b) { const double _t = (a); (a) = (b); (b) = _t; }
__host__ __device__ inline void ldlt6_solve(double A[36], const double rhs[6], double x[6]) {
  int piv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = fabs(A[k * 6 + k]);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const double v = fabs(A[i * 6 + i]);
      if (v > best) {
        best = v;
        p = i;
      }
    }
    piv[k] = p;
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
      if (p == i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) NG_SWAP(A[k * 6 + j], A[i * 6 + j]);
#pragma unroll
        for (int j = 0; j < 6; ++j) NG_SWAP(A[j * 6 + k], A[j * 6 + i]);
      }
    const double dk = A[k * 6 + k];
    if (dk != 0.0 && isfinite(dk)) {
#pragma unroll
      for (int i = k + 1; i < 6; ++i) A[i * 6 + k] /= dk;
#pragma unroll
      for (int i = k + 1; i < 6; ++i)
#pragma unroll
        for (int j = k + 1; j <= i; ++j) {
          A[i * 6 + j] -= A[i * 6 + k] * dk * A[j * 6 + k];
          A[j * 6 + i] = A[i * 6 + j];
        }
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = rhs[i];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
      if (piv[k] == i) NG_SWAP(y[k], y[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) y[i] -= A[i * 6 + j] * y[j];
  const double tol = 2.2250738585072014e-308;
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = (fabs(A[i * 6 + i]) > tol) ? y[i] / A[i * 6 + i] : 0.0;
#pragma unroll
  for (int i = 5; i >= 0; --i)
#pragma unroll
    for (int j = i + 1; j < 6; ++j) y[i] -= A[j * 6 + i] * y[j];
#pragma unroll
  for (int k = 5; k >= 0; --k)
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
      if (piv[k] == i) NG_SWAP(y[k], y[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}
#undef NG_SWAP

}  // namespace ngk
