// =============================================================================
// ngicp_oracle.cpp — CPU restatement of DLO's NanoGICP hot path.
//
// *** TEST INFRASTRUCTURE ONLY ***  This file is the *checker*.  Only tests/,
// __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load the
// library built from it.  The product path (direct_lidar_odometry_amd/csrc)
// never links, loads or calls anything in oracle/.
//
// PARITY STATUS: the reference ships no tests / golden vectors for this path
//  and its GICP layer needs Eigen + PCL which are absent
// here, so the GICP math below is "parity unpinned" by the reference's own
// fixtures.  It is pinned instead by (i) the *real* reference kd-tree compiled
// from /root/reference (oracle/ref_nanoflann.cpp -> oracle/_ref/), against
// which the kd-tree below is checked index-for-index, and (ii) an independent
// float64 numpy/scipy restatement (oracle/numpy_model.py).
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/include/nano_gicp/).  No Eigen/PCL: small fixed-size linear
// algebra is written out by hand in FP64 exactly where the reference is FP64,
// and in FP32 (no FMA contraction: build with -ffp-contract=off) exactly where
// the reference is FP32 (point transform for the NN query and squared
// distances).
// =============================================================================
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ----------------------------------------------------------------------------
// Strided float xyz view of a cloud (PointXYZI = 8 floats per point, dlo.h:50)
// ----------------------------------------------------------------------------
struct CloudView {
  const float* p = nullptr;
  size_t n = 0;
  size_t stride = 0;  // in floats
  inline float at(size_t i, int d) const { return p[i * stride + d]; }
};

// ----------------------------------------------------------------------------
// pcl::transformPointCloud with a float matrix (impl/lsq_registration_impl.hpp:114; src/dlo/odom.cc:484,971-974).  PCL is a
// third-party dependency that is NOT under /root/reference (PCL >= 1.10, unpinned: README.md:27) and not installed here: this
// restates pcl/common/impl/transforms.hpp (detail::Transformer<float>::se3) FROM MEMORY -> parity unpinned.
//   order 1 (SSE2 path, what an x86-64 build of PCL >= 1.9 executes): x*c0 + (y*c1 + (z*c2 + c3)), c_j = matrix columns
//   order 0 (scalar fallback):                                        ((m_r0*x + m_r1*y) + m_r2*z) + m_r3
// m is column-major.  Float arithmetic, un-fused (-ffp-contract=off).
// ----------------------------------------------------------------------------
inline void transform_point_pcl(const float* m, float x, float y, float z, int sse_order, float* out3) {
  for (int r = 0; r < 3; ++r) {
    if (sse_order) {
      const float a = z * m[8 + r] + m[12 + r];
      const float b = y * m[4 + r] + a;
      out3[r] = x * m[0 + r] + b;
    } else {
      out3[r] = ((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r];
    }
  }
}

// ----------------------------------------------------------------------------
// kd-tree: restatement of the single-index static tree the reference uses
// (nanoflann.hpp:100-102,114 => L2 metric on 3 floats, int index, leaf 100).
// Flat node array instead of a pooled pointer tree; same split rule, same
// permutation, same traversal order => same results including tie order.
// ----------------------------------------------------------------------------
struct KdNode {
  int child_lo = -1, child_hi = -1;  // -1/-1 => leaf
  int first = 0, last = 0;           // leaf: range in perm
  int axis = 0;                      // inner: split axis
  float div_lo = 0.f, div_hi = 0.f;  // inner: tight max of low child / tight min of high child
};

struct Box3 {
  float lo[3], hi[3];
};

class KdTree {
 public:
  static constexpr int kLeafMax = 100;  // nanoflann.hpp:114

  void build(const CloudView& c) {
    cloud_ = c;
    nodes_.clear();
    perm_.resize(c.n);
    for (size_t i = 0; i < c.n; ++i) perm_[i] = (int)i;  // impl/nanoflann_impl.hpp:1316-1323
    root_ = -1;
    if (c.n == 0) return;
    // impl/nanoflann_impl.hpp:1325-1346 — full scan bounding box
    for (int d = 0; d < 3; ++d) root_box_.lo[d] = root_box_.hi[d] = c.at(0, d);
    for (size_t k = 1; k < c.n; ++k)
      for (int d = 0; d < 3; ++d) {
        float v = c.at(k, d);
        if (v < root_box_.lo[d]) root_box_.lo[d] = v;
        if (v > root_box_.hi[d]) root_box_.hi[d] = v;
      }
    nodes_.reserve(c.n / 25 + 16);
    root_ = divide(0, (int)c.n, root_box_);
  }

  size_t size() const { return cloud_.n; }
  const CloudView& cloud() const { return cloud_; }

  // Exact k-NN, ascending, strict-'>' insertion (impl/nanoflann_impl.hpp:184-211,
  // 1230-1250, 1355-1418; nanoflann.hpp:141-152).  Returns number found.
  int knn(const float q[3], int k, int* out_idx, float* out_d2) const {
    if (cloud_.n == 0 || k <= 0) return 0;
    Result r{out_idx, out_d2, k, 0};
    out_d2[k - 1] = std::numeric_limits<float>::max();
    float dists[3] = {0.f, 0.f, 0.f};
    float d2 = 0.f;
    // impl/nanoflann_impl.hpp:1014-1031
    for (int d = 0; d < 3; ++d) {
      if (q[d] < root_box_.lo[d]) {
        dists[d] = (q[d] - root_box_.lo[d]) * (q[d] - root_box_.lo[d]);
        d2 += dists[d];
      }
      if (q[d] > root_box_.hi[d]) {
        dists[d] = (q[d] - root_box_.hi[d]) * (q[d] - root_box_.hi[d]);
        d2 += dists[d];
      }
    }
    descend(r, q, root_, d2, dists);
    return r.count;
  }

 private:
  struct Result {
    int* idx;
    float* d2;
    int cap;
    int count;
    inline float worst() const { return d2[cap - 1]; }
    inline void add(float dist, int index) {  // impl/nanoflann_impl.hpp:184-211
      int i = count;
      while (i > 0 && d2[i - 1] > dist) {
        if (i < cap) {
          d2[i] = d2[i - 1];
          idx[i] = idx[i - 1];
        }
        --i;
      }
      if (i < cap) {
        d2[i] = dist;
        idx[i] = index;
      }
      if (count < cap) ++count;
    }
  };

  inline float coord(int permuted, int d) const { return cloud_.at((size_t)permuted, d); }

  void minmax(const int* ind, int count, int d, float& mn, float& mx) const {
    mn = mx = coord(ind[0], d);
    for (int i = 1; i < count; ++i) {
      float v = coord(ind[i], d);
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
  }

  // Two-phase Hoare partition (impl/nanoflann_impl.hpp:976-1012): afterwards
  // [0,lim1) < cut, [lim1,lim2) == cut, [lim2,count) > cut.
  template <class Pred>
  int hoare(int* ind, int count, int start, Pred goes_left) const {
    int l = start, r = count - 1;
    for (;;) {
      while (l <= r && goes_left(ind[l])) ++l;
      while (r && l <= r && !goes_left(ind[r])) --r;
      if (l > r || !r) break;
      std::swap(ind[l], ind[r]);
      ++l;
      --r;
    }
    return l;
  }

  // impl/nanoflann_impl.hpp:919-965
  void middle_split(int* ind, int count, const Box3& box, int& split_idx, int& axis, float& cut) const {
    const float kEps = 0.00001f;
    float max_span = box.hi[0] - box.lo[0];
    for (int d = 1; d < 3; ++d) max_span = std::max(max_span, box.hi[d] - box.lo[d]);
    float best_spread = -1.f;
    axis = 0;
    for (int d = 0; d < 3; ++d) {
      float span = box.hi[d] - box.lo[d];
      if (span > (1 - kEps) * max_span) {
        float mn, mx;
        minmax(ind, count, d, mn, mx);
        float spread = mx - mn;
        if (spread > best_spread) {
          axis = d;
          best_spread = spread;
        }
      }
    }
    float mid = (box.lo[axis] + box.hi[axis]) / 2;
    float mn, mx;
    minmax(ind, count, axis, mn, mx);
    cut = mid < mn ? mn : (mid > mx ? mx : mid);
    const int a = axis;
    const float c = cut;
    int lim1 = hoare(ind, count, 0, [&](int p) { return coord(p, a) < c; });
    int lim2 = hoare(ind, count, lim1, [&](int p) { return coord(p, a) <= c; });
    if (lim1 > count / 2)
      split_idx = lim1;
    else if (lim2 < count / 2)
      split_idx = lim2;
    else
      split_idx = count / 2;
  }

  // impl/nanoflann_impl.hpp:867-917.  `box` is in/out: on return it is the
  // tight box of the subtree.
  int divide(int left, int right, Box3& box) {
    int me = (int)nodes_.size();
    nodes_.emplace_back();
    if (right - left <= kLeafMax) {
      nodes_[me].first = left;
      nodes_[me].last = right;
      for (int d = 0; d < 3; ++d) box.lo[d] = box.hi[d] = coord(perm_[left], d);
      for (int k = left + 1; k < right; ++k)
        for (int d = 0; d < 3; ++d) {
          float v = coord(perm_[k], d);
          if (box.lo[d] > v) box.lo[d] = v;
          if (box.hi[d] < v) box.hi[d] = v;
        }
      return me;
    }
    int idx, axis;
    float cut;
    middle_split(perm_.data() + left, right - left, box, idx, axis, cut);
    Box3 lbox = box, rbox = box;
    lbox.hi[axis] = cut;
    rbox.lo[axis] = cut;
    int lo_child = divide(left, left + idx, lbox);
    int hi_child = divide(left + idx, right, rbox);
    KdNode& nd = nodes_[me];
    nd.axis = axis;
    nd.child_lo = lo_child;
    nd.child_hi = hi_child;
    nd.div_lo = lbox.hi[axis];
    nd.div_hi = rbox.lo[axis];
    for (int d = 0; d < 3; ++d) {
      box.lo[d] = std::min(lbox.lo[d], rbox.lo[d]);
      box.hi[d] = std::max(lbox.hi[d], rbox.hi[d]);
    }
    return me;
  }

  // impl/nanoflann_impl.hpp:1355-1418 with epsError = 1 (nanoflann.hpp:150)
  void descend(Result& r, const float q[3], int node, float mind2, float dists[3]) const {
    const KdNode& nd = nodes_[node];
    if (nd.child_lo < 0) {
      float worst = r.worst();  // snapshot at leaf entry, as the reference does
      for (int i = nd.first; i < nd.last; ++i) {
        int p = perm_[i];
        // impl/nanoflann_impl.hpp:441-449: result += diff*diff for x,y,z in order
        float dx = q[0] - coord(p, 0), dy = q[1] - coord(p, 1), dz = q[2] - coord(p, 2);
        float d = 0.f;
        d += dx * dx;
        d += dy * dy;
        d += dz * dz;
        if (d < worst) r.add(d, p);
      }
      return;
    }
    int a = nd.axis;
    float v = q[a];
    float diff1 = v - nd.div_lo, diff2 = v - nd.div_hi;
    int best, other;
    float cut;
    if (diff1 + diff2 < 0) {
      best = nd.child_lo;
      other = nd.child_hi;
      cut = (v - nd.div_hi) * (v - nd.div_hi);
    } else {
      best = nd.child_hi;
      other = nd.child_lo;
      cut = (v - nd.div_lo) * (v - nd.div_lo);
    }
    descend(r, q, best, mind2, dists);
    float keep = dists[a];
    mind2 = mind2 + cut - keep;
    dists[a] = cut;
    if (mind2 * 1.0f <= r.worst()) descend(r, q, other, mind2, dists);
    dists[a] = keep;
  }

  CloudView cloud_;
  std::vector<int> perm_;
  std::vector<KdNode> nodes_;
  Box3 root_box_{};
  int root_ = -1;
};

// ----------------------------------------------------------------------------
// Small FP64 linear algebra (stand-ins for the Eigen calls the reference makes)
// ----------------------------------------------------------------------------
struct Iso3 {  // Eigen::Isometry3d restated: column-major 4x4 affine, R | t
  double R[3][3];
  double t[3];
};

static Iso3 iso_identity() {
  Iso3 x{};
  for (int i = 0; i < 3; ++i) x.R[i][i] = 1.0;
  return x;
}

static Iso3 iso_mul(const Iso3& a, const Iso3& b) {  // a * b
  Iso3 c{};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a.R[i][k] * b.R[k][j];
      c.R[i][j] = s;
    }
    double s = 0;
    for (int k = 0; k < 3; ++k) s += a.R[i][k] * b.t[k];
    c.t[i] = s + a.t[i];
  }
  return c;
}

static Iso3 iso_from_colmajor_f(const float m[16]) {  // Isometry3d(guess.cast<double>())
  Iso3 x{};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) x.R[r][c] = (double)m[c * 4 + r];
    x.t[r] = (double)m[12 + r];
  }
  return x;
}
static Iso3 iso_from_colmajor_d(const double m[16]) {
  Iso3 x{};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) x.R[r][c] = m[c * 4 + r];
    x.t[r] = m[12 + r];
  }
  return x;
}
static void iso_to_colmajor_f(const Iso3& x, float m[16]) {  // x0.cast<float>().matrix()
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) m[c * 4 + r] = (float)x.R[r][c];
    m[12 + r] = (float)x.t[r];
    m[r * 4 + 3] = 0.f;
  }
  m[15] = 1.f;
}

// gicp/so3.hpp:99-118 followed by Eigen's Quaternion::toRotationMatrix()
static void so3_exp_matrix(const double w[3], double R[3][3]) {
  double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    double theta = std::sqrt(theta_sq);
    double half = 0.5 * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  const double qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0][0] = 1 - (tyy + tzz);
  R[0][1] = txy - twz;
  R[0][2] = txz + twy;
  R[1][0] = txy + twz;
  R[1][1] = 1 - (txx + tzz);
  R[1][2] = tyz - twx;
  R[2][0] = txz - twy;
  R[2][1] = tyz + twx;
  R[2][2] = 1 - (txx + tyy);
}

// 3x3 symmetric inverse by cofactors (the 3x3 block of the reference's
// Matrix4d::inverse() at impl/nano_gicp_impl.hpp:205-209).
static void inv3_sym(const double a[6] /*xx xy xz yy yz zz*/, double o[6]) {
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

static void inv3_general(const double m[3][3], double o[3][3]) {
  const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  const double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  const double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  const double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
  const double id = 1.0 / det;
  o[0][0] = c00 * id;
  o[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
  o[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
  o[1][0] = c01 * id;
  o[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
  o[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  o[2][0] = c02 * id;
  o[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
  o[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (stands in for
// JacobiSVD<Matrix3d> at impl/nano_gicp_impl.hpp:332: for a symmetric PSD
// matrix U == V == eigenvectors and the singular values are the eigenvalues).
// Classical (Rutishauser) rotation update: a_pp -= t a_pq, a_qq += t a_pq.
// Output: w descending, V columns are the matching unit eigenvectors.
static void eig3_sym(const double a_in[3][3], double w[3], double V[3][3]) {
  double a[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = a_in[i][j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  static const int PQ[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};  // (p, q, r): pivot pair and the third index
  for (int sweep = 0; sweep < 24; ++sweep) {
    double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    double diag = std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-22 * diag) break;
    for (int e = 0; e < 3; ++e) {
      const int p = PQ[e][0], q = PQ[e][1], r = PQ[e][2];
      const double apq = a[p][q];
      if (apq == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      a[p][p] = a[p][p] - t * apq;
      a[q][q] = a[q][q] + t * apq;
      a[p][q] = a[q][p] = 0.0;
      const double rp = a[r][p], rq = a[r][q];
      a[r][p] = a[p][r] = c * rp - s * rq;
      a[r][q] = a[q][r] = s * rp + c * rq;
      for (int k = 0; k < 3; ++k) {
        const double vp = V[k][p], vq = V[k][q];
        V[k][p] = c * vp - s * vq;
        V[k][q] = s * vp + c * vq;
      }
    }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  std::sort(order, order + 3, [&](int x, int y) { return std::fabs(d[x]) > std::fabs(d[y]); });
  double Vs[3][3];
  for (int j = 0; j < 3; ++j) {
    w[j] = d[order[j]];
    for (int i = 0; i < 3; ++i) Vs[i][j] = V[i][order[j]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

// 6x6 LDLT with diagonal pivoting, solve A x = rhs (stands in for
// Eigen::LDLT<Matrix6d>::solve at impl/lsq_registration_impl.hpp:147,172).
// Zero pivots contribute zero to the solution, as Eigen's solve does.
static void ldlt6_solve(const double A_in[6][6], const double rhs[6], double x[6]) {
  double A[6][6];
  std::memcpy(A, A_in, sizeof(A));
  int piv[6];
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(A[i][i]) > best) {
        best = std::fabs(A[i][i]);
        p = i;
      }
    piv[k] = p;
    if (p != k) {
      for (int j = 0; j < 6; ++j) std::swap(A[k][j], A[p][j]);
      for (int i = 0; i < 6; ++i) std::swap(A[i][k], A[i][p]);
    }
    double dk = A[k][k];
    if (dk != 0.0 && std::isfinite(dk)) {
      for (int i = k + 1; i < 6; ++i) A[i][k] /= dk;
      for (int i = k + 1; i < 6; ++i)
        for (int j = k + 1; j <= i; ++j) {
          A[i][j] -= A[i][k] * dk * A[j][k];
          A[j][i] = A[i][j];
        }
    }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) y[i] = rhs[i];
  for (int k = 0; k < 6; ++k) std::swap(y[k], y[piv[k]]);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < 6; ++i) y[i] = (std::fabs(A[i][i]) > tol) ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; --i)
    for (int j = i + 1; j < 6; ++j) y[i] -= A[j][i] * y[j];
  for (int k = 5; k >= 0; --k) std::swap(y[k], y[piv[k]]);
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// ----------------------------------------------------------------------------
// Covariances  (impl/nano_gicp_impl.hpp:300-357)
// ----------------------------------------------------------------------------
enum Regularization { REG_NONE = 0, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS };  // gicp_settings.hpp:47

using Mat4 = double[16];  // column-major 4x4, as Eigen::Matrix4d

static inline void store_cov4(double* m /*16*/, const double c[3][3]) {
  std::memset(m, 0, sizeof(double) * 16);
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) m[cc * 4 + r] = c[r][cc];
}

static int calc_covariances(const CloudView& cloud, const KdTree& tree, int k, int reg, double* covs, int threads) {
  const long n = (long)cloud.n;
  if (k <= 0) return -1;
  if ((size_t)k > cloud.n) return -2;  // undefined in the reference; explicit error here
#pragma omp parallel for num_threads(threads) schedule(guided, 8)
  for (long i = 0; i < n; ++i) {
    std::vector<int> idx(k);
    std::vector<float> d2(k);
    float q[3] = {cloud.at(i, 0), cloud.at(i, 1), cloud.at(i, 2)};
    tree.knn(q, k, idx.data(), d2.data());
    // :315-321 — 4xk double matrix, mean-centre, C = X X^T / k
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < k; ++j)
      for (int d = 0; d < 3; ++d) mean[d] += (double)cloud.at(idx[j], d);
    for (int d = 0; d < 3; ++d) mean[d] /= (double)k;
    double C[3][3] = {{0}};
    for (int j = 0; j < k; ++j) {
      double v[3];
      for (int d = 0; d < 3; ++d) v[d] = (double)cloud.at(idx[j], d) - mean[d];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r][c] += v[r] * v[c];
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) C[r][c] /= (double)k;

    double out[3][3];
    if (reg == REG_NONE) {  // :323-324
      std::memcpy(out, C, sizeof(C));
    } else if (reg == REG_FROBENIUS) {  // :325-330
      double Cl[3][3], Ci[3][3];
      std::memcpy(Cl, C, sizeof(C));
      for (int d = 0; d < 3; ++d) Cl[d][d] += 1e-3;
      inv3_general(Cl, Ci);
      double nrm = 0;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) nrm += Ci[r][c] * Ci[r][c];
      nrm = std::sqrt(nrm);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Ci[r][c] /= nrm;
      inv3_general(Ci, out);
    } else {  // :331-353
      double w[3], V[3][3], vals[3];
      eig3_sym(C, w, V);
      if (reg == REG_PLANE) {
        vals[0] = 1;
        vals[1] = 1;
        vals[2] = 1e-3;
      } else if (reg == REG_MIN_EIG) {
        for (int d = 0; d < 3; ++d) vals[d] = std::max(std::fabs(w[d]), 1e-3);
      } else {
        double mx = std::max(std::fabs(w[0]), std::max(std::fabs(w[1]), std::fabs(w[2])));
        for (int d = 0; d < 3; ++d) vals[d] = std::max(std::fabs(w[d]) / mx, 1e-3);
      }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double s = 0;
          for (int e = 0; e < 3; ++e) s += V[r][e] * vals[e] * V[c][e];
          out[r][c] = s;
        }
    }
    store_cov4(covs + (size_t)i * 16, out);
  }
  return 0;
}

// ----------------------------------------------------------------------------
// The registration object: restates NanoGICP + LsqRegistration state & methods
// ----------------------------------------------------------------------------
struct Gicp {
  // parameters (defaults: impl/nano_gicp_impl.hpp:50-64, impl/lsq_registration_impl.hpp:50-63,
  // PCL Registration defaults: max_iterations_ overwritten to 64, corr dist FLT_MAX)
  int num_threads = 1;
  int k = 20;
  int reg = REG_PLANE;
  double corr_dist = (double)std::numeric_limits<float>::max();
  int max_iterations = 64;
  double rot_eps = 2e-3;
  double trans_eps = 5e-4;
  int optimizer = 1;  // 0 GN, 1 LM
  int lm_max_iterations = 10;
  double lm_init_lambda_factor = 1e-9;
  bool debug_print = false;

  // clouds are *copied* by the oracle (the reference holds shared_ptrs)
  std::vector<float> src_pts, tgt_pts;  // packed xyz1, 4 floats / point
  uint64_t src_id = 0, tgt_id = 0;      // pointer identity stand-in
  std::shared_ptr<KdTree> src_tree, tgt_tree;
  std::vector<double> src_covs, tgt_covs;  // n x 16
  std::vector<double> mahal;               // n x 16 (only 3x3 block non-zero)
  std::vector<int> corr;
  std::vector<float> sqd;

  // results
  double lm_lambda = -1.0;
  bool converged = false;
  int nr_iterations = 0;
  float final_T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double final_hessian[36];
  std::vector<double> trace;  // per LM trial: iter, trial, y0, yi, rho, lambda, |d|, accepted

  Gicp() {
#ifdef _OPENMP
    num_threads = omp_get_max_threads();
#endif
    std::memset(final_hessian, 0, sizeof(final_hessian));
    for (int i = 0; i < 6; ++i) final_hessian[i * 6 + i] = 1.0;  // impl/lsq_registration_impl.hpp:62
  }

  size_t n_src() const { return src_pts.size() / 4; }
  size_t n_tgt() const { return tgt_pts.size() / 4; }
  CloudView src_view() const { return CloudView{src_pts.data(), n_src(), 4}; }
  CloudView tgt_view() const { return CloudView{tgt_pts.data(), n_tgt(), 4}; }

  static void pack(const float* xyz, size_t n, size_t stride_floats, std::vector<float>& out) {
    out.resize(n * 4);
    for (size_t i = 0; i < n; ++i) {
      out[i * 4 + 0] = xyz[i * stride_floats + 0];
      out[i * 4 + 1] = xyz[i * stride_floats + 1];
      out[i * 4 + 2] = xyz[i * stride_floats + 2];
      out[i * 4 + 3] = 1.0f;  // PointXYZI data[3] == 1
    }
  }

  // impl/nano_gicp_impl.hpp:174-211
  void update_correspondences(const Iso3& T) {
    const long n = (long)n_src();
    corr.resize(n);
    sqd.resize(n);
    mahal.resize((size_t)n * 16);
    float Tf[3][4];  // trans.cast<float>()
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) Tf[r][c] = (float)T.R[r][c];
      Tf[r][3] = (float)T.t[r];
    }
    const double gate = corr_dist * corr_dist;
    const CloudView sv = src_view();
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
    for (long i = 0; i < n; ++i) {
      const float x = sv.at(i, 0), y = sv.at(i, 1), z = sv.at(i, 2), w = sv.at(i, 3);
      float q[3];
      // Eigen 4x4 * 4-vector in float: column-scaled sum, left to right, no FMA
      for (int r = 0; r < 3; ++r) q[r] = ((Tf[r][0] * x + Tf[r][1] * y) + Tf[r][2] * z) + Tf[r][3] * w;
      int j = -1;
      float d2 = 0.f;
      tgt_tree->knn(q, 1, &j, &d2);
      sqd[i] = d2;
      corr[i] = ((double)d2 < gate) ? j : -1;
      if (corr[i] < 0) continue;
      const double* A = &src_covs[(size_t)i * 16];
      const double* B = &tgt_covs[(size_t)corr[i] * 16];
      // RCR = C_B + R C_A R^T  (3x3 block; the 4th row/col of the 4x4 only carries the (3,3)=1 trick)
      double RA[3][3], S[6];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double s = 0;
          for (int e = 0; e < 3; ++e) s += T.R[r][e] * A[c * 4 + e];
          RA[r][c] = s;
        }
      int t = 0;
      for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) {
          double s = 0;
          for (int e = 0; e < 3; ++e) s += RA[r][e] * T.R[c][e];
          S[t++] = B[c * 4 + r] + s;
        }
      double Mi[6];
      inv3_sym(S, Mi);
      double* M = &mahal[(size_t)i * 16];
      std::memset(M, 0, sizeof(double) * 16);
      M[0] = Mi[0];
      M[1] = M[4] = Mi[1];
      M[2] = M[8] = Mi[2];
      M[5] = Mi[3];
      M[6] = M[9] = Mi[4];
      M[10] = Mi[5];
    }
  }

  // impl/nano_gicp_impl.hpp:214-270 (H,b may be null) and :273-296
  double accumulate(const Iso3& T, double* H36, double* b6) {
    const long n = (long)n_src();
    const int nt = std::max(1, num_threads);
    std::vector<double> Hs((size_t)nt * 36, 0.0), bs((size_t)nt * 6, 0.0);
    double sum = 0.0;
    const CloudView sv = src_view();
    const CloudView tv = tgt_view();
    const bool want = (H36 && b6);
#pragma omp parallel for num_threads(nt) reduction(+ : sum) schedule(guided, 8)
    for (long i = 0; i < n; ++i) {
      int j = corr[i];
      if (j < 0) continue;
      double a[3] = {(double)sv.at(i, 0), (double)sv.at(i, 1), (double)sv.at(i, 2)};
      double bpt[3] = {(double)tv.at(j, 0), (double)tv.at(j, 1), (double)tv.at(j, 2)};
      double ta[3], e[3];
      for (int r = 0; r < 3; ++r) ta[r] = T.R[r][0] * a[0] + T.R[r][1] * a[1] + T.R[r][2] * a[2] + T.t[r];
      for (int r = 0; r < 3; ++r) e[r] = bpt[r] - ta[r];
      const double* M = &mahal[(size_t)i * 16];
      double Me[3];
      for (int r = 0; r < 3; ++r) Me[r] = M[0 * 4 + r] * e[0] + M[1 * 4 + r] * e[1] + M[2 * 4 + r] * e[2];
      sum += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
      if (!want) continue;
      // J = [ skew(ta) | -I ]  (3x6)
      double J[3][6] = {{0, -ta[2], ta[1], -1, 0, 0}, {ta[2], 0, -ta[0], 0, -1, 0}, {-ta[1], ta[0], 0, 0, 0, -1}};
      double MJ[3][6];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 6; ++c) MJ[r][c] = M[0 * 4 + r] * J[0][c] + M[1 * 4 + r] * J[1][c] + M[2 * 4 + r] * J[2][c];
#ifdef _OPENMP
      int tid = omp_get_thread_num();
#else
      int tid = 0;
#endif
      double* Ht = &Hs[(size_t)tid * 36];
      double* bt = &bs[(size_t)tid * 6];
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) Ht[c * 6 + r] += J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c];
        bt[r] += J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2];
      }
    }
    if (want) {
      std::memset(H36, 0, sizeof(double) * 36);
      std::memset(b6, 0, sizeof(double) * 6);
      for (int t = 0; t < nt; ++t) {
        for (int i = 0; i < 36; ++i) H36[i] += Hs[(size_t)t * 36 + i];
        for (int i = 0; i < 6; ++i) b6[i] += bs[(size_t)t * 6 + i];
      }
    }
    return sum;
  }

  double linearize(const Iso3& T, double* H36, double* b6) {
    update_correspondences(T);
    return accumulate(T, H36, b6);
  }
  double compute_error(const Iso3& T) { return accumulate(T, nullptr, nullptr); }

  // impl/lsq_registration_impl.hpp:118-127
  bool is_converged(const Iso3& d) const {
    double rmax = 0, tmax = 0;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) rmax = std::max(rmax, 1.0 / rot_eps * std::fabs(d.R[r][c] - (r == c ? 1.0 : 0.0)));
      tmax = std::max(tmax, 1.0 / trans_eps * std::fabs(d.t[r]));
    }
    return std::max(rmax, tmax) < 1;
  }

  static Iso3 delta_from(const double d[6]) {
    Iso3 delta = iso_identity();
    so3_exp_matrix(d, delta.R);
    delta.t[0] = d[3];
    delta.t[1] = d[4];
    delta.t[2] = d[5];
    return delta;
  }

  // impl/lsq_registration_impl.hpp:142-158
  bool step_gn(Iso3& x0, Iso3& delta) {
    double H[36], b[6], Hm[6][6], rhs[6], d[6];
    linearize(x0, H, b);
    for (int r = 0; r < 6; ++r) {
      for (int c = 0; c < 6; ++c) Hm[r][c] = H[c * 6 + r];
      rhs[r] = -b[r];
    }
    ldlt6_solve(Hm, rhs, d);
    delta = delta_from(d);
    x0 = iso_mul(delta, x0);
    std::memcpy(final_hessian, H, sizeof(H));
    return true;
  }

  // impl/lsq_registration_impl.hpp:161-208
  bool step_lm(Iso3& x0, Iso3& delta, int outer) {
    double H[36], b[6];
    double y0 = linearize(x0, H, b);
    if (lm_lambda < 0.0) {
      double m = 0;
      for (int i = 0; i < 6; ++i) m = std::max(m, std::fabs(H[i * 6 + i]));
      lm_lambda = lm_init_lambda_factor * m;
    }
    double nu = 2.0;
    for (int i = 0; i < lm_max_iterations; ++i) {
      double Hm[6][6], rhs[6], d[6];
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) Hm[r][c] = H[c * 6 + r] + (r == c ? lm_lambda : 0.0);
        rhs[r] = -b[r];
      }
      ldlt6_solve(Hm, rhs, d);
      delta = delta_from(d);
      Iso3 xi = iso_mul(delta, x0);
      double yi = compute_error(xi);
      double den = 0, dn = 0;
      for (int r = 0; r < 6; ++r) {
        den += d[r] * (lm_lambda * d[r] - b[r]);
        dn += d[r] * d[r];
      }
      double rho = (y0 - yi) / den;
      const bool rejected = (rho < 0);
      trace.insert(trace.end(), {(double)outer, (double)i, y0, yi, rho, lm_lambda, std::sqrt(dn), rejected ? 0.0 : 1.0});
      if (debug_print) std::printf("%5d %5d %15g %15g %15g %15g %15g\n", outer, i, y0, yi, rho, lm_lambda, std::sqrt(dn));
      if (rejected) {
        if (is_converged(delta)) return true;
        lm_lambda = nu * lm_lambda;
        nu = 2 * nu;
        continue;
      }
      x0 = xi;
      lm_lambda = lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
      std::memcpy(final_hessian, H, sizeof(H));
      return true;
    }
    return false;
  }

  // impl/nano_gicp_impl.hpp:162-171 + impl/lsq_registration_impl.hpp:89-115 (+ PCL align() prologue)
  int align(const float guess[16], float* aligned_xyz, size_t aligned_stride_floats) {
    if (n_src() == 0 || n_tgt() == 0 || !tgt_tree) return -3;
    converged = false;
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(final_T, I, sizeof(I));
    trace.clear();
    if (src_covs.size() != n_src() * 16) {
      if (!src_tree) return -4;
      src_covs.resize(n_src() * 16);
      int rc = calc_covariances(src_view(), *src_tree, k, reg, src_covs.data(), num_threads);
      if (rc) return rc;
    }
    if (tgt_covs.size() != n_tgt() * 16) {
      tgt_covs.resize(n_tgt() * 16);
      int rc = calc_covariances(tgt_view(), *tgt_tree, k, reg, tgt_covs.data(), num_threads);
      if (rc) return rc;
    }
    Iso3 x0 = iso_from_colmajor_f(guess);
    lm_lambda = -1.0;
    converged = false;
    nr_iterations = 0;
    for (int i = 0; i < max_iterations && !converged; ++i) {
      nr_iterations = i;
      Iso3 delta = iso_identity();
      bool ok = optimizer == 0 ? step_gn(x0, delta) : step_lm(x0, delta, i);
      if (!ok) break;  // "lm not converged!!"
      converged = is_converged(delta);
    }
    iso_to_colmajor_f(x0, final_T);
    if (aligned_xyz) {  // pcl::transformPointCloud with the float matrix
      const CloudView sv = src_view();
      const float* m = final_T;
      for (size_t i = 0; i < sv.n; ++i) {
        float x = sv.at(i, 0), y = sv.at(i, 1), z = sv.at(i, 2);
        transform_point_pcl(m, x, y, z, 1, aligned_xyz + i * aligned_stride_floats);
      }
    }
    return 0;
  }
};

}  // namespace

// =============================================================================
// C ABI of the oracle (ctypes-friendly).  Names are orc_* to keep them apart
// from the product's ngicp_* symbols.
// =============================================================================
extern "C" {

struct orc_tree {
  std::vector<float> pts;
  KdTree tree;
};

orc_tree* orc_tree_build(const float* xyz, size_t n, size_t stride_floats) {
  orc_tree* t = new orc_tree;
  t->pts.resize(n * 4);
  for (size_t i = 0; i < n; ++i) {
    for (int d = 0; d < 3; ++d) t->pts[i * 4 + d] = xyz[i * stride_floats + d];
    t->pts[i * 4 + 3] = 1.f;
  }
  t->tree.build(CloudView{t->pts.data(), n, 4});
  return t;
}
void orc_tree_free(orc_tree* t) { delete t; }

int orc_tree_knn(const orc_tree* t, const float* queries, size_t nq, size_t qstride_floats, int k, int* idx, float* d2, int threads) {
  if (!t || k <= 0) return -1;
  if (threads <= 0) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(guided, 8)
  for (long i = 0; i < (long)nq; ++i) {
    int found = t->tree.knn(queries + (size_t)i * qstride_floats, k, idx + (size_t)i * k, d2 + (size_t)i * k);
    for (int j = found; j < k; ++j) {
      idx[(size_t)i * k + j] = -1;
      d2[(size_t)i * k + j] = std::numeric_limits<float>::infinity();
    }
  }
  return 0;
}

int orc_covariances(const float* xyz, size_t n, size_t stride_floats, int k, int reg, double* covs_n16, int threads) {
  orc_tree* t = orc_tree_build(xyz, n, stride_floats);
  int rc = calc_covariances(CloudView{t->pts.data(), n, 4}, t->tree, k, reg, covs_n16, threads <= 0 ? 1 : threads);
  orc_tree_free(t);
  return rc;
}

void orc_so3_exp(const double w[3], double R_rowmajor[9]) {
  double R[3][3];
  so3_exp_matrix(w, R);
  std::memcpy(R_rowmajor, R, sizeof(R));
}
void orc_ldlt6_solve(const double A_rowmajor[36], const double rhs[6], double x[6]) {
  double A[6][6];
  std::memcpy(A, A_rowmajor, sizeof(A));
  ldlt6_solve(A, rhs, x);
}
void orc_eig3_sym(const double A_rowmajor[9], double w[3], double V_rowmajor[9]) {
  double A[3][3], V[3][3];
  std::memcpy(A, A_rowmajor, sizeof(A));
  eig3_sym(A, w, V);
  std::memcpy(V_rowmajor, V, sizeof(V));
}

typedef struct Gicp orc_gicp;

orc_gicp* orc_gicp_create() { return new Gicp; }
void orc_gicp_destroy(orc_gicp* g) { delete g; }

int orc_gicp_set_params(orc_gicp* g, int k, double max_corr_dist, int max_iter, double trans_eps, double rot_eps, int optimizer, int lm_max_iter,
                        double lm_init_lambda_factor, int regularization, int num_threads) {
  g->k = k;
  g->corr_dist = max_corr_dist;
  g->max_iterations = max_iter;
  g->trans_eps = trans_eps;
  g->rot_eps = rot_eps;
  g->optimizer = optimizer;
  g->lm_max_iterations = lm_max_iter;
  g->lm_init_lambda_factor = lm_init_lambda_factor;
  g->reg = regularization;
  if (num_threads > 0) g->num_threads = num_threads;
#ifdef _OPENMP
  else
    g->num_threads = omp_get_max_threads();  // impl/nano_gicp_impl.hpp:70-78
#endif
  return 0;
}
int orc_gicp_num_threads(const orc_gicp* g) { return g->num_threads; }
void orc_gicp_set_debug(orc_gicp* g, int on) { g->debug_print = on != 0; }

// impl/nano_gicp_impl.hpp:121-129 — identity early-out, build tree, clear covs
int orc_gicp_set_source(orc_gicp* g, const float* xyz, size_t n, size_t stride_floats, uint64_t identity) {
  if (identity != 0 && identity == g->src_id) return 0;
  Gicp::pack(xyz, n, stride_floats, g->src_pts);
  g->src_id = identity;
  g->src_tree = std::make_shared<KdTree>();
  g->src_tree->build(g->src_view());
  g->src_covs.clear();
  return 0;
}
// impl/nano_gicp_impl.hpp:113-118 — no tree build, covs untouched
int orc_gicp_register_source(orc_gicp* g, const float* xyz, size_t n, size_t stride_floats, uint64_t identity) {
  if (identity != 0 && identity == g->src_id) return 0;
  Gicp::pack(xyz, n, stride_floats, g->src_pts);
  g->src_id = identity;
  return 0;
}
// impl/nano_gicp_impl.hpp:132-139
int orc_gicp_set_target(orc_gicp* g, const float* xyz, size_t n, size_t stride_floats, uint64_t identity) {
  if (identity != 0 && identity == g->tgt_id) return 0;
  Gicp::pack(xyz, n, stride_floats, g->tgt_pts);
  g->tgt_id = identity;
  g->tgt_tree = std::make_shared<KdTree>();
  g->tgt_tree->build(g->tgt_view());
  g->tgt_covs.clear();
  return 0;
}
// src/dlo/odom.cc:525 — `gicp.source_kdtree_ = gicp_s2s.source_kdtree_`
// NB: the shared tree keeps looking at the *donor's* point buffer, as the reference's does.
int orc_gicp_share_source_index(orc_gicp* dst, orc_gicp* src) {
  dst->src_tree = src->src_tree;
  return 0;
}
int orc_gicp_copy_source_covs(orc_gicp* dst, const orc_gicp* src) {  // odom.cc:815
  dst->src_covs = src->src_covs;
  return 0;
}
int orc_gicp_clear_source_covs(orc_gicp* g) {  // odom.cc:526
  g->src_covs.clear();
  return 0;
}
int orc_gicp_compute_source_covs(orc_gicp* g) {  // impl/nano_gicp_impl.hpp:152-154,300-357
  if (!g->src_tree || g->src_tree->size() != g->n_src() || g->src_tree->cloud().p != g->src_pts.data()) {
    g->src_tree = std::make_shared<KdTree>();  // :304-306 re-set tree input when its cloud differs
    g->src_tree->build(g->src_view());
  }
  g->src_covs.resize(g->n_src() * 16);
  return calc_covariances(g->src_view(), *g->src_tree, g->k, g->reg, g->src_covs.data(), g->num_threads);
}
int orc_gicp_compute_target_covs(orc_gicp* g) {
  if (!g->tgt_tree) return -4;
  g->tgt_covs.resize(g->n_tgt() * 16);
  return calc_covariances(g->tgt_view(), *g->tgt_tree, g->k, g->reg, g->tgt_covs.data(), g->num_threads);
}
size_t orc_gicp_source_covs_size(const orc_gicp* g) { return g->src_covs.size() / 16; }
size_t orc_gicp_target_covs_size(const orc_gicp* g) { return g->tgt_covs.size() / 16; }
int orc_gicp_get_source_covs(const orc_gicp* g, double* out) {
  std::memcpy(out, g->src_covs.data(), g->src_covs.size() * sizeof(double));
  return 0;
}
int orc_gicp_get_target_covs(const orc_gicp* g, double* out) {
  std::memcpy(out, g->tgt_covs.data(), g->tgt_covs.size() * sizeof(double));
  return 0;
}
int orc_gicp_set_source_covs(orc_gicp* g, const double* in, size_t n) {  // impl/nano_gicp_impl.hpp:142-144
  g->src_covs.assign(in, in + n * 16);
  return 0;
}
int orc_gicp_set_target_covs(orc_gicp* g, const double* in, size_t n) {  // :147-149
  g->tgt_covs.assign(in, in + n * 16);
  return 0;
}
// impl/nano_gicp_impl.hpp:91-98
int orc_gicp_swap_source_target(orc_gicp* g) {
  g->src_pts.swap(g->tgt_pts);
  std::swap(g->src_id, g->tgt_id);
  g->src_tree.swap(g->tgt_tree);
  g->src_covs.swap(g->tgt_covs);
  g->corr.clear();
  g->sqd.clear();
  return 0;
}

int orc_gicp_align(orc_gicp* g, const float guess_colmajor[16], float T_out_colmajor[16], int* converged, int* nr_iterations,
                   double final_hessian_colmajor[36], float* aligned_xyz_or_null, size_t aligned_stride_floats) {
  int rc = g->align(guess_colmajor, aligned_xyz_or_null, aligned_stride_floats);
  if (T_out_colmajor) std::memcpy(T_out_colmajor, g->final_T, sizeof(g->final_T));
  if (converged) *converged = g->converged ? 1 : 0;
  if (nr_iterations) *nr_iterations = g->nr_iterations;
  if (final_hessian_colmajor) std::memcpy(final_hessian_colmajor, g->final_hessian, sizeof(g->final_hessian));
  return rc;
}

// test hooks: linearize / compute_error at a given FP64 pose
int orc_gicp_linearize(orc_gicp* g, const double T_colmajor[16], double H_colmajor[36], double b[6], double* err) {
  if (g->src_covs.size() != g->n_src() * 16 || g->tgt_covs.size() != g->n_tgt() * 16) return -5;
  Iso3 T = iso_from_colmajor_d(T_colmajor);
  *err = g->linearize(T, H_colmajor, b);
  return 0;
}
int orc_gicp_compute_error(orc_gicp* g, const double T_colmajor[16], double* err) {
  if (g->corr.size() != g->n_src()) return -6;
  Iso3 T = iso_from_colmajor_d(T_colmajor);
  *err = g->compute_error(T);
  return 0;
}
int orc_gicp_get_correspondences(const orc_gicp* g, int* corr, float* sqd) {
  if (corr) std::memcpy(corr, g->corr.data(), g->corr.size() * sizeof(int));
  if (sqd) std::memcpy(sqd, g->sqd.data(), g->sqd.size() * sizeof(float));
  return (int)g->corr.size();
}
int orc_gicp_get_mahalanobis(const orc_gicp* g, double* out_n16) {
  std::memcpy(out_n16, g->mahal.data(), g->mahal.size() * sizeof(double));
  return 0;
}
size_t orc_gicp_trace_rows(const orc_gicp* g) { return g->trace.size() / 8; }
int orc_gicp_get_trace(const orc_gicp* g, double* out) {
  std::memcpy(out, g->trace.data(), g->trace.size() * sizeof(double));
  return 0;
}
double orc_gicp_lambda(const orc_gicp* g) { return g->lm_lambda; }

// ----------------------------------------------------------------------------
// "filters": pcl::removeNaNFromPointCloud, pcl::CropBox (negative) and pcl::VoxelGrid as DLO uses them
// (/root/reference/src/dlo/odom.cc:443-465 with :122-127; src/dlo/map.cc:100-131).  PCL is NOT under /root/reference
// (PCL >= 1.10, unpinned: README.md:27) and not installed here: restated FROM MEMORY of pcl/filters/filter.hpp,
// pcl/filters/impl/crop_box.hpp and pcl/filters/impl/voxel_grid.hpp -> parity unpinned.
//   in : n strided points (xyz at float 0..2, intensity at float `ioff`, or ioff < 0: none)
//   out: packed {x, y, z, intensity} (4 floats per point), returns the number of points written
// VoxelGrid: inverse_leaf = 1 / leaf (float); bounding box of the input -> min_b = floor(min * inv), div_b = max_b - min_b + 1;
// voxel index (i - min_b.x) + (j - min_b.y) div.x + (k - min_b.z) div.x div.y; points sorted by index; per voxel the
// centroid of all fields = float sums / count (pcl::CentroidPoint), output in ascending index.  PCL's std::sort leaves the
// order inside a voxel unspecified; here the sort is stable (input order), which only affects the last bits of the sums.
// If the index would overflow int32 PCL warns and returns the input unchanged.
// ----------------------------------------------------------------------------
size_t orc_filter_cloud(const float* pts, size_t n, size_t stride_floats, long ioff, int remove_nan, float crop_half, float leaf, float* out_xyzi) {
  std::vector<float> cur;  // packed survivors
  cur.reserve(n * 4);
  const bool voxel = leaf > 0.f;
  for (size_t i = 0; i < n; ++i) {
    const float* p = pts + i * stride_floats;
    const float x = p[0], y = p[1], z = p[2], it = ioff >= 0 ? p[ioff] : 0.f;
    if ((remove_nan || voxel) && !(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) continue;
    if (crop_half > 0.f && !(x < -crop_half || y < -crop_half || z < -crop_half || x > crop_half || y > crop_half || z > crop_half)) continue;
    cur.push_back(x); cur.push_back(y); cur.push_back(z); cur.push_back(it);
  }
  size_t m = cur.size() / 4;
  if (!voxel || m == 0) {
    std::memcpy(out_xyzi, cur.data(), cur.size() * sizeof(float));
    return m;
  }
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (size_t i = 0; i < m; ++i)
    for (int d = 0; d < 3; ++d) {
      mn[d] = std::min(mn[d], cur[i * 4 + d]);
      mx[d] = std::max(mx[d], cur[i * 4 + d]);
    }
  const float inv = 1.0f / leaf;
  int min_b[3], div[3];
  long long cells = 1;
  for (int d = 0; d < 3; ++d) {
    min_b[d] = (int)std::floor(mn[d] * inv);
    div[d] = (int)std::floor(mx[d] * inv) - min_b[d] + 1;
    cells *= (long long)div[d];
    if (cells > (long long)std::numeric_limits<int>::max()) {  // "Leaf size is too small for the input dataset": output = input
      std::memcpy(out_xyzi, cur.data(), cur.size() * sizeof(float));
      return m;
    }
  }
  std::vector<std::pair<unsigned int, unsigned int>> idx(m);
  for (size_t i = 0; i < m; ++i) {
    const int a = (int)std::floor(cur[i * 4 + 0] * inv) - min_b[0], b = (int)std::floor(cur[i * 4 + 1] * inv) - min_b[1], c = (int)std::floor(cur[i * 4 + 2] * inv) - min_b[2];
    idx[i] = {(unsigned int)(a + b * div[0] + c * div[0] * div[1]), (unsigned int)i};
  }
  std::stable_sort(idx.begin(), idx.end(), [](const std::pair<unsigned int, unsigned int>& l, const std::pair<unsigned int, unsigned int>& r) { return l.first < r.first; });
  size_t nv = 0;
  for (size_t s0 = 0; s0 < m;) {
    size_t e0 = s0;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    while (e0 < m && idx[e0].first == idx[s0].first) {
      const float* p = &cur[(size_t)idx[e0].second * 4];
      sx += p[0]; sy += p[1]; sz += p[2]; si += p[3];
      ++e0;
    }
    const float cnt = (float)(e0 - s0);
    out_xyzi[nv * 4 + 0] = sx / cnt; out_xyzi[nv * 4 + 1] = sy / cnt; out_xyzi[nv * 4 + 2] = sz / cnt; out_xyzi[nv * 4 + 3] = si / cnt;
    ++nv;
    s0 = e0;
  }
  return nv;
}

// pcl::transformPointCloud restated (see transform_point_pcl): strided xyz in, packed xyz out
int orc_transform_cloud(const float* xyz, size_t n, size_t stride_floats, const float T_colmajor[16], int sse_order, float* out_xyz) {
  for (size_t i = 0; i < n; ++i) transform_point_pcl(T_colmajor, xyz[i * stride_floats], xyz[i * stride_floats + 1], xyz[i * stride_floats + 2], sse_order, out_xyz + i * 3);
  return 0;
}

}  // extern "C"
