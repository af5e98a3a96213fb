// nano_gicp::NanoGICP<PointSource, PointTarget> — header-only drop-in for DLO's scan-matching class, running
// on an MI355X through the C ABI of include/ngicp.h (hand-written HIP kernels; no CPU fallback).
//
// It reproduces the public surface the reference exposes and DLO uses
// (/root/reference/include/nano_gicp/nano_gicp.hpp:79-125, lsq_registration.hpp:75-89 and the
// pcl::Registration entry points called at /root/reference/src/dlo/odom.cc:100-120,479-480,498-500,519-526,
// 803-837): same method names, argument meaning and failure behaviour (nothing throws on the normal path;
// failure = hasConverged() == false plus a line on stderr), including the two PUBLIC DATA MEMBERS the odometry
// node touches directly:
//     gicp.source_kdtree_ = gicp_s2s.source_kdtree_;    (odom.cc:525)  -> shares the device-resident index
//     gicp.source_covs_.clear();                        (odom.cc:526)
//     gicp.source_covs_   = gicp_s2s.source_covs_;      (odom.cc:815)  -> device-to-device, no host round trip
//
// With PCL and Eigen installed the class uses pcl::PointCloud / Eigen::Matrix4f; without them (this repo's
// build container) it uses the layout-compatible stand-ins of pcl_compat.hpp.
#pragma once
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../ngicp.h"

#if defined(__has_include)
#if __has_include(<pcl/point_cloud.h>) && __has_include(<Eigen/Core>)
#define NGICP_HAVE_PCL 1
#endif
#endif

#ifdef NGICP_HAVE_PCL
#include <Eigen/Core>
#include <Eigen/StdVector>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/search/kdtree.h>
namespace nano_gicp {
namespace types {
template <class P> using Cloud = pcl::PointCloud<P>;
using Matrix4f = Eigen::Matrix4f;
using Matrix4d = Eigen::Matrix4d;
using Matrix6d = Eigen::Matrix<double, 6, 6>;
using CovVector = std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>;
}  // namespace types
}  // namespace nano_gicp
#else
#include "pcl_compat.hpp"
namespace pcl {
using PointXYZI = ngicp_compat::PointXYZI;
template <class P> using PointCloud = ngicp_compat::PointCloud<P>;
template <class S, class T, class Scalar = float> using Registration = ngicp_compat::Registration<S, T, Scalar>;
namespace search {
template <class P> using KdTree = ngicp_compat::SearchKdTree<P>;
}
}  // namespace pcl
namespace Eigen {  // the two spellings DLO's members use (include/dlo/odom.h:104,133; src/dlo/odom.cc:809)
using Matrix4f = ngicp_compat::Matrix4f;
using Matrix4d = ngicp_compat::Matrix4d;
template <class T> using aligned_allocator = std::allocator<T>;
}  // namespace Eigen
namespace nano_gicp {
namespace types {
template <class P> using Cloud = ngicp_compat::PointCloud<P>;
using Matrix4f = ngicp_compat::Matrix4f;
using Matrix4d = ngicp_compat::Matrix4d;
using Matrix6d = ngicp_compat::Matrix6d;
using CovVector = std::vector<ngicp_compat::Matrix4d>;
}  // namespace types
}  // namespace nano_gicp
#endif

namespace nano_gicp {

enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS };  // gicp/gicp_settings.hpp:47
enum class LSQ_OPTIMIZER_TYPE { GaussNewton, LevenbergMarquardt };                         // lsq_registration.hpp:54

template <typename PointSource, typename PointTarget>
class NanoGICP {
 public:
  using Scalar = float;
  using Matrix4 = types::Matrix4f;
  using PointCloudSource = types::Cloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = types::Cloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using CovVector = types::CovVector;
#ifdef NGICP_HAVE_PCL
  using KdTreeReciprocal = pcl::search::KdTree<PointSource>;  // as pcl::Registration<PointSource, PointTarget>
#else
  using KdTreeReciprocal = typename pcl::Registration<PointSource, PointTarget, Scalar>::KdTreeReciprocal;
#endif
  using KdTreeReciprocalPtr = typename KdTreeReciprocal::Ptr;

  // ---- proxies for the public data members DLO assigns to ----
  struct IndexRef {  // stands for std::shared_ptr<nanoflann::KdTreeFLANN<PointSource>> source_kdtree_
    NanoGICP* owner = nullptr;
    IndexRef& operator=(const IndexRef& o) {
      if (owner && o.owner && owner != o.owner) owner->check(ngicp_share_source_index(owner->h_, o.owner->h_), "share_source_index");
      return *this;
    }
  };
  struct CovRef {  // stands for std::vector<Eigen::Matrix4d> source_covs_ / target_covs_
    NanoGICP* owner = nullptr;
    bool source = true;
    void clear() { owner->check(source ? ngicp_clear_source_covs(owner->h_) : ngicp_clear_target_covs(owner->h_), "clear_covs"); }
    size_t size() const {
      size_t n = 0;
      if (source) ngicp_source_covs_size(owner->h_, &n); else ngicp_target_covs_size(owner->h_, &n);
      return n;
    }
    CovRef& operator=(const CovRef& o) {
      if (this == &o) return *this;
      if (source && o.source) {
        owner->check(ngicp_copy_source_covs(owner->h_, o.owner->h_), "copy_source_covs");  // stays on the device
      } else {
        CovVector tmp = o.owner->fetch(o.source);
        *this = tmp;
      }
      return *this;
    }
    CovRef& operator=(const CovVector& v) {
      owner->check(source ? ngicp_set_source_covs(owner->h_, reinterpret_cast<const double*>(v.data()), v.size())
                          : ngicp_set_target_covs(owner->h_, reinterpret_cast<const double*>(v.data()), v.size()), "set_covs");
      return *this;
    }
    operator CovVector() const { return owner->fetch(source); }
  };

  explicit NanoGICP(int device = 0) {
    int rc = ngicp_create(device, &h_);
    if (rc != NGICP_OK) {
      std::fprintf(stderr, "[NanoGICP] cannot create the GPU engine: %s\n", ngicp_last_error(nullptr));
      h_ = nullptr;
    }
    source_kdtree_.owner = target_kdtree_.owner = this;
    source_covs_.owner = target_covs_.owner = this;
    source_covs_.source = true;
    target_covs_.source = false;
    set_identity(final_transformation_);
    final_hessian_ = types::Matrix6d::Identity();
  }
  virtual ~NanoGICP() { ngicp_destroy(h_); }
  NanoGICP(const NanoGICP&) = delete;
  NanoGICP& operator=(const NanoGICP&) = delete;

  bool valid() const { return h_ != nullptr; }

  // ---- NanoGICP / LsqRegistration setters (impl/nano_gicp_impl.hpp:70-88, impl/lsq_registration_impl.hpp:69-81) ----
  void setNumThreads(int n) { num_threads_ = n; push(); }
  void setCorrespondenceRandomness(int k) { k_correspondences_ = k; push(); }
  void setRegularizationMethod(RegularizationMethod m) { regularization_method_ = m; push(); }
  void setRotationEpsilon(double e) { rotation_epsilon_ = e; push(); }
  void setInitialLambdaFactor(double f) { lm_init_lambda_factor_ = f; push(); }
  void setDebugPrint(bool on) { lm_debug_print_ = on; }
  // ---- pcl::Registration setters DLO calls (odom.cc:101-120) ----
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; push(); }
  void setMaximumIterations(int n) { max_iterations_ = n; push(); }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; push(); }
  void setEuclideanFitnessEpsilon(double) {}               // accepted, never read by nano_gicp
  void setRANSACIterations(int) {}                         // "
  void setRANSACOutlierRejectionThreshold(double) {}       // "
  // PCL's own FLANN trees are never used by nano_gicp (DLO hands over null pointers with force_no_recompute = true so that
  // initCompute() does not build them, odom.cc:116-120): any pointer type is accepted - pcl::search::KdTree<P>::Ptr with
  // real PCL, the stand-in of pcl_compat.hpp without - and dropped.
  template <class TreePtr> void setSearchMethodSource(const TreePtr&, bool /*force_no_recompute*/ = false) {}
  template <class TreePtr> void setSearchMethodTarget(const TreePtr&, bool /*force_no_recompute*/ = false) {}
  void setLSQType(LSQ_OPTIMIZER_TYPE t) { lsq_optimizer_type_ = t; push(); }

  // ---- clouds (impl/nano_gicp_impl.hpp:101-139) ----
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) {
    if (input_ == cloud) return;
    input_ = cloud;
    check(ngicp_set_source(h_, xyz(cloud), cloud->size(), sizeof(PointSource), id(cloud)), "setInputSource");
  }
  virtual void registerInputSource(const PointCloudSourceConstPtr& cloud) {
    if (input_ == cloud) return;
    input_ = cloud;
    check(ngicp_register_source(h_, xyz(cloud), cloud->size(), sizeof(PointSource), id(cloud)), "registerInputSource");
  }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) {
    if (target_ == cloud && !device_target_) return;
    target_ = cloud;
    device_target_ = false;
    check(ngicp_set_target(h_, xyz(cloud), cloud->size(), sizeof(PointTarget), id(cloud)), "setInputTarget");
  }
  virtual void clearSource() { input_.reset(); check(ngicp_clear_source(h_), "clearSource"); }
  virtual void clearTarget() { target_.reset(); device_target_ = false; check(ngicp_clear_target(h_), "clearTarget"); }
  virtual void swapSourceAndTarget() {  // :91-98
    if (device_target_) {  // a device-assembled submap has no host cloud to become the input: materialise it first
      std::fprintf(stderr, "[NanoGICP] swapSourceAndTarget(): the target is a device-resident submap; not swapped\n");
      return;
    }
    input_.swap(target_);
    check(ngicp_swap_source_target(h_), "swapSourceAndTarget");
  }
  PointCloudSourceConstPtr getInputSource() const { return input_; }
  PointCloudTargetConstPtr getInputTarget() const { return target_; }

  // ---- covariances (:142-159, nano_gicp.hpp:100-106) ----
  virtual void setSourceCovariances(const CovVector& c) { source_covs_ = c; }
  virtual void setTargetCovariances(const CovVector& c) { target_covs_ = c; }
  virtual bool calculateSourceCovariances() { return check(ngicp_compute_source_covs(h_), "calculateSourceCovariances"); }
  virtual bool calculateTargetCovariances() { return check(ngicp_compute_target_covs(h_), "calculateTargetCovariances"); }
  const CovVector& getSourceCovariances() const { cache_src_ = fetch(true); return cache_src_; }
  const CovVector& getTargetCovariances() const { cache_tgt_ = fetch(false); return cache_tgt_; }

  // ---- pcl::Registration::align: copies input into output, identity final transform, then
  //      computeTransformation; output[i].data[3] = 1 ----
  void align(PointCloudSource& output) { Matrix4 I; set_identity(I); align(output, I); }
  void align(PointCloudSource& output, const Matrix4& guess) {
    converged_ = false;
    set_identity(final_transformation_);
    if (!h_ || !input_ || !(target_ || device_target_)) {  // PCL's initCompute() fails silently
      std::fprintf(stderr, "[NanoGICP] align(): no input source/target\n");
      return;
    }
    output = *input_;
    int conv = 0, nit = 0;
    std::vector<float> xyz_out(input_->size() * 3);
    int rc = ngicp_align(h_, guess.data(), final_transformation_.data(), &conv, &nit, final_hessian_.data(), xyz_out.data(), 12);
    converged_ = conv != 0;
    nr_iterations_ = nit;
    if (rc != NGICP_OK) {
      std::fprintf(stderr, "[NanoGICP] align(): %s\n", ngicp_last_error(h_));
      return;
    }
    if (lm_debug_print_) print_lm_table();
    for (size_t i = 0; i < output.size(); ++i) {  // transformed xyz; the other fields stay as in the input
      output.points[i].data[0] = xyz_out[i * 3 + 0];
      output.points[i].data[1] = xyz_out[i * 3 + 1];
      output.points[i].data[2] = xyz_out[i * 3 + 2];
      output.points[i].data[3] = 1.0f;
    }
  }
  // same, without materialising the aligned cloud (DLO never reads it: odom.cc:799-837)
  void alignPoseOnly(const Matrix4& guess) {
    converged_ = false;
    set_identity(final_transformation_);
    if (!h_ || !input_ || !(target_ || device_target_)) return;
    int conv = 0, nit = 0;
    int rc = ngicp_align(h_, guess.data(), final_transformation_.data(), &conv, &nit, final_hessian_.data(), nullptr, 0);
    converged_ = conv != 0;
    nr_iterations_ = nit;
    if (rc != NGICP_OK) std::fprintf(stderr, "[NanoGICP] align(): %s\n", ngicp_last_error(h_));
    else if (lm_debug_print_) print_lm_table();
  }
  // setDebugPrint(true): the reference prints a banner per align() and one row per LM trial while it optimises
  // (impl/lsq_registration_impl.hpp:95-99,183-189: boost::format "%5d %15g %15g %15g %15g %15g %5c" of i, y0, yi, rho, lambda,
  // |delta|, 'x' when rho > 0, a header in front of trial 0).  The loop runs on the device here, so the same table is printed
  // from the engine's trace once align() has returned.
  void print_lm_table() const {
    size_t n = 0;
    if (ngicp_get_lm_trace(h_, nullptr, 0, &n) != NGICP_OK) return;
    std::vector<double> rows(n * 8);
    if (n && ngicp_get_lm_trace(h_, rows.data(), n, &n) != NGICP_OK) return;
    std::printf("********************************************\n***************** optimize *****************\n********************************************\n");
    for (size_t r = 0; r < n; ++r) {
      const double* q = &rows[r * 8];  // {outer iteration, trial, y0, yi, rho, lambda, |delta|, accepted}
      if ((int)q[1] == 0) std::printf("--- LM optimization ---\n%5s %15s %15s %15s %15s %15s %5s\n", "i", "y0", "yi", "rho", "lambda", "|delta|", "dec");
      std::printf("%5d %15g %15g %15g %15g %15g %5c\n", (int)q[1], q[2], q[3], q[4], q[5], q[6], q[4] > 0.0 ? 'x' : ' ');
    }
    std::fflush(stdout);
  }
  Matrix4 getFinalTransformation() const { return final_transformation_; }
  bool hasConverged() const { return converged_; }
  const types::Matrix6d& getFinalHessian() const { return final_hessian_; }
  int getNrIterations() const { return nr_iterations_; }
  ngicp_t* handle() { return h_; }

  // ---- extensions with no reference counterpart (include/ngicp.h "keyframe store", "rigid transform"): the keyframes and the
  //      submap stay on the GPU.  In DLO they replace odom.cc:1174 (`keyframe_normals.push_back(gicp_s2s.getSourceCovariances())`)
  //      and odom.cc:830-833 (`setInputTarget(submap_cloud); setTargetCovariances(submap_normals)`); see INTEGRATION.md ----
  // adopt `producer`'s current source cloud + covariances as the next keyframe; returns its index (== DLO's keyframes.size() - 1)
  int addKeyframe(NanoGICP& producer) {
    int id = -1;
    check(ngicp_keyframe_add(h_, producer.h_, &id), "addKeyframe");
    return id;
  }
  // the same with the keyframe cloud = producer's current source transformed by T on the device (odom.cc:971-974 + 1166-1174)
  int addKeyframeTransformed(NanoGICP& producer, const Matrix4& T) {
    int id = -1;
    check(ngicp_keyframe_add_transformed(h_, producer.h_, T.data(), &id), "addKeyframeTransformed");
    return id;
  }
  // the same with the submap voxel filter in between (DLO's shipped configuration: vf_submap_use_, odom.cc:1160-1163)
  int addKeyframeTransformedFiltered(NanoGICP& producer, const Matrix4& T, float leaf) {
    int id = -1;
    check(ngicp_keyframe_add_transformed_filtered(h_, producer.h_, T.data(), leaf, &id), "addKeyframeTransformedFiltered");
    return id;
  }
  size_t numKeyframes() const { size_t n = 0; ngicp_keyframe_count(h_, &n); return n; }
  // target := concatenation of the given keyframes (cloud + covariances), assembled and indexed on the device; a call with
  // the id list of the current submap is a no-op.  Returns true when the target was rebuilt.
  bool setSubmapKeyframes(const std::vector<int>& ids) {
    int changed = 0;
    if (check(ngicp_submap_set(h_, ids.data(), ids.size(), &changed), "setSubmapKeyframes")) {
      target_.reset();
      device_target_ = true;
    }
    return changed != 0;
  }
  // dlo::OdomNode::preprocessPoints (odom.cc:443-465): removeNaN -> CropBox(negative, +-crop_size) -> VoxelGrid(voxel_res), each
  // optional (crop_size / voxel_res <= 0: off), on the GPU; `cloud` is replaced by the filtered cloud (x, y, z, intensity;
  // data[3] = 1).  With set_as_source the filtered cloud, still on the device, also becomes this instance's input source.
  void preprocessPoints(PointCloudSource& cloud, bool remove_nan, float crop_size, float voxel_res, bool set_as_source = false) {
    if (!h_ || cloud.empty()) return;
    std::vector<float> out(cloud.size() * 4);
    size_t m = 0;
    const long ioff = (long)((const char*)&cloud.points[0].intensity - (const char*)cloud.points[0].data);
    if (!check(ngicp_preprocess_scan(h_, cloud.points[0].data, cloud.size(), sizeof(PointSource), ioff, remove_nan ? 1 : 0, crop_size, voxel_res, out.data(),
                                     cloud.size(), &m), "preprocessPoints"))
      return;
    cloud.resize(m);
    for (size_t i = 0; i < m; ++i) {
      PointSource& p = cloud.points[i];
      p.data[0] = out[i * 4 + 0]; p.data[1] = out[i * 4 + 1]; p.data[2] = out[i * 4 + 2]; p.data[3] = 1.0f;
      p.intensity = out[i * 4 + 3];
    }
    if (set_as_source) check(ngicp_set_source_preprocessed(h_, 0), "set_source_preprocessed");
  }
  // dlo::MapNode (map.cc:100-131): `*dlo_map += *keyframe` and `voxelgrid.filter(*dlo_map)` on a device-resident map
  void mapAdd(const PointCloudSource& keyframe) {
    if (!h_ || keyframe.empty()) return;
    const long ioff = (long)((const char*)&keyframe.points[0].intensity - (const char*)keyframe.points[0].data);
    check(ngicp_map_add(h_, keyframe.points[0].data, keyframe.size(), sizeof(PointSource), ioff), "mapAdd");
  }
  size_t mapVoxelFilter(float leaf) { size_t m = 0; check(ngicp_map_voxel_filter(h_, leaf, &m), "mapVoxelFilter"); return m; }
  void mapGet(PointCloudSource& out) {
    size_t n = 0;
    ngicp_map_size(h_, &n);
    std::vector<float> buf(n * 4);
    if (n && !check(ngicp_map_get(h_, buf.data(), n), "mapGet")) return;
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
      PointSource& p = out.points[i];
      p.data[0] = buf[i * 4 + 0]; p.data[1] = buf[i * 4 + 1]; p.data[2] = buf[i * 4 + 2]; p.data[3] = 1.0f;
      p.intensity = buf[i * 4 + 3];
    }
  }
  // pcl::transformPointCloud(*getInputSource(), out, T) computed from the device-resident source (odom.cc:971-974)
  void transformSource(PointCloudSource& out, const Matrix4& T) {
    if (!h_ || !input_) return;
    out = *input_;
    std::vector<float> xyz_out(input_->size() * 3);
    if (!check(ngicp_transform_source(h_, T.data(), xyz_out.data(), 12), "transformSource")) return;
    for (size_t i = 0; i < out.size(); ++i) {
      out.points[i].data[0] = xyz_out[i * 3 + 0];
      out.points[i].data[1] = xyz_out[i * 3 + 1];
      out.points[i].data[2] = xyz_out[i * 3 + 2];
    }
  }

 public:  // the reference's public data members (nano_gicp.hpp:120-125)
  IndexRef source_kdtree_, target_kdtree_;
  CovRef source_covs_, target_covs_;

 protected:
  CovVector fetch(bool source) const {
    size_t n = 0;
    if (source) ngicp_source_covs_size(h_, &n); else ngicp_target_covs_size(h_, &n);
    CovVector v(n);
    static_assert(sizeof(typename CovVector::value_type) == 16 * sizeof(double), "Matrix4d must be 16 doubles");
    if (n) (source ? ngicp_get_source_covs : ngicp_get_target_covs)(h_, reinterpret_cast<double*>(v.data()));
    return v;
  }
  bool check(int rc, const char* what) const {
    if (rc != NGICP_OK) std::fprintf(stderr, "[NanoGICP] %s: %s\n", what, h_ ? ngicp_last_error(h_) : "no engine");
    return rc == NGICP_OK;
  }
  void push() {
    if (!h_) return;
    check(ngicp_set_params(h_, k_correspondences_, corr_dist_threshold_, max_iterations_, transformation_epsilon_, rotation_epsilon_,
                           lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::LevenbergMarquardt ? 1 : 0, lm_max_iterations_, lm_init_lambda_factor_,
                           (int)regularization_method_, num_threads_), "set_params");
  }
  template <class CloudPtr> static const float* xyz(const CloudPtr& c) { return c->size() ? c->points[0].data : nullptr; }
  template <class CloudPtr> static uint64_t id(const CloudPtr& c) { return (uint64_t)(uintptr_t)c.get(); }
  static void set_identity(Matrix4& m) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m(r, c) = r == c ? 1.f : 0.f; }

  ngicp_t* h_ = nullptr;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  bool device_target_ = false;  // the target is a device-assembled submap (setSubmapKeyframes): no host cloud behind it
  // defaults: impl/nano_gicp_impl.hpp:50-64, impl/lsq_registration_impl.hpp:50-63
  int num_threads_ = 0;
  int k_correspondences_ = 20;
  RegularizationMethod regularization_method_ = RegularizationMethod::PLANE;
  double corr_dist_threshold_ = (double)std::numeric_limits<float>::max();
  int max_iterations_ = 64;
  double transformation_epsilon_ = 5e-4;
  double rotation_epsilon_ = 2e-3;
  LSQ_OPTIMIZER_TYPE lsq_optimizer_type_ = LSQ_OPTIMIZER_TYPE::LevenbergMarquardt;
  int lm_max_iterations_ = 10;
  double lm_init_lambda_factor_ = 1e-9;
  bool lm_debug_print_ = false;
  Matrix4 final_transformation_;
  types::Matrix6d final_hessian_;
  bool converged_ = false;
  int nr_iterations_ = 0;
  mutable CovVector cache_src_, cache_tgt_;
};

}  // namespace nano_gicp
