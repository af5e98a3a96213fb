// Minimal stand-ins for the few PCL / Eigen types the NanoGICP public API mentions, used ONLY when the real
// libraries are not installed (this build container has neither PCL nor Eigen).  With PCL/Eigen present the
// shim (nano_gicp.hpp) uses the real pcl::PointCloud / pcl::PointXYZI / Eigen::Matrix4f instead.
// Layouts follow the real types: PointXYZI is 32 bytes (float data[4] with data[3] = 1, intensity padded to
// 16 bytes; /root/reference/include/dlo/dlo.h:50), matrices are column-major like Eigen's default.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace ngicp_compat {

struct alignas(16) PointXYZI {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union {
    float data_c[4];
    struct {
      float intensity;
    };
  };
  PointXYZI() : data{0.f, 0.f, 0.f, 1.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
  PointXYZI(float x_, float y_, float z_, float i_ = 0.f) : data{x_, y_, z_, 1.f}, data_c{i_, 0.f, 0.f, 0.f} {}
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");

template <class PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 1;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(std::size_t n) { points.resize(n); width = (std::uint32_t)n; }
  void clear() { points.clear(); width = 0; }
  PointT& at(std::size_t i) { return points.at(i); }
  const PointT& at(std::size_t i) const { return points.at(i); }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
  void push_back(const PointT& p) { points.push_back(p); width = (std::uint32_t)points.size(); }
};

// column-major fixed-size matrix with the handful of members the call sites use
template <class T, int N>
struct Matrix {
  T m[N * N];
  Matrix() { for (int i = 0; i < N * N; ++i) m[i] = T(0); }
  static Matrix Identity() { Matrix r; for (int i = 0; i < N; ++i) r.m[i * N + i] = T(1); return r; }
  static Matrix Zero() { return Matrix(); }
  T& operator()(int r, int c) { return m[c * N + r]; }
  const T& operator()(int r, int c) const { return m[c * N + r]; }
  T* data() { return m; }
  const T* data() const { return m; }
  template <class U> Matrix<U, N> cast() const { Matrix<U, N> r; for (int i = 0; i < N * N; ++i) r.m[i] = (U)m[i]; return r; }
  Matrix operator*(const Matrix& o) const {
    Matrix r;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        T s = T(0);
        for (int k = 0; k < N; ++k) s += (*this)(i, k) * o(k, j);
        r(i, j) = s;
      }
    return r;
  }
};
using Matrix4f = Matrix<float, 4>;
using Matrix4d = Matrix<double, 4>;
using Matrix6d = Matrix<double, 6>;

// pcl::search::KdTree<PointT> — only ever passed around as a (null) pointer (src/dlo/odom.cc:116-120)
template <class PointT>
struct SearchKdTree {
  using Ptr = std::shared_ptr<SearchKdTree<PointT>>;
  using ConstPtr = std::shared_ptr<const SearchKdTree<PointT>>;
};

// pcl::Registration<PointSource, PointTarget, Scalar> — the names DLO's call sites spell out through the base class
// (src/dlo/odom.cc:116: `pcl::Registration<PointType, PointType>::KdTreeReciprocalPtr temp;`) and the typedefs the
// reference class pulls from it (include/nano_gicp/nano_gicp.hpp:60-70).
template <class PointSource, class PointTarget, class Scalar = float>
struct Registration {
  using Matrix4 = Matrix<Scalar, 4>;
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudTarget = PointCloud<PointTarget>;
  using KdTree = SearchKdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using KdTreeReciprocal = SearchKdTree<PointSource>;
  using KdTreeReciprocalPtr = typename KdTreeReciprocal::Ptr;
};

}  // namespace ngicp_compat
