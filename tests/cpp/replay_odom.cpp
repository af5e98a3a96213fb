// Replays the NanoGICP call sequence of DLO's odometry node through the header-only shim, on clouds read from raw float32
// files, and prints the resulting transforms.  The pytest driver compares them with the CPU oracle run through the same sequence.
// Every statement that touches `gicp` / `gicp_s2s` is spelled the way the odometry node spells it (`this->` members with the
// node's member names; /root/reference/src/dlo/odom.cc line numbers in the comments); the ROS / filter / hull code between
// those statements is not part of the boundary and is replaced by the few lines of glue this file needs (load, transform,
// keyframe every second scan).
//   usage: replay_odom [--device-keyframes] <n_scans> <scan0.bin> <scan1.bin> ...   (each file: N x 3 float32)
// --device-keyframes: keyframes and the submap stay on the GPU (NanoGICP::addKeyframe / setSubmapKeyframes, the fast path that
// replaces odom.cc:1174 and :830-833); without it the submap cloud and its covariances take the reference's host route.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "nano_gicp/nano_gicp.hpp"

using PointType = pcl::PointXYZI;  // include/dlo/dlo.h:50

namespace dlo {

struct OdomNode {
  // include/dlo/odom.h:100-133 (the members the call sites below name)
  pcl::PointCloud<PointType>::Ptr current_scan, current_scan_t, target_cloud, keyframe_cloud, submap_cloud;
  std::vector<pcl::PointCloud<PointType>::Ptr> keyframes;
  std::vector<std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>> keyframe_normals;
  std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>> submap_normals;
  std::vector<int> submap_kf_idx_curr, submap_kf_idx_prev;
  bool submap_hasChanged = true;
  nano_gicp::NanoGICP<PointType, PointType> gicp_s2s;
  nano_gicp::NanoGICP<PointType, PointType> gicp;
  Eigen::Matrix4f T, T_s2s, T_s2s_prev;
  bool imu_use_ = false;
  Eigen::Matrix4f imu_SE3;
  bool device_keyframes = false;

  // cfg/params.yaml:54-71
  int gicps2s_k_correspondences_ = 10, gicps2m_k_correspondences_ = 20;
  double gicps2s_max_corr_dist_ = 1.0, gicps2m_max_corr_dist_ = 0.5;
  int gicps2s_max_iter_ = 32, gicps2m_max_iter_ = 32;
  double gicps2s_transformation_ep_ = 0.01, gicps2m_transformation_ep_ = 0.01;
  double gicps2s_euclidean_fitness_ep_ = 0.01, gicps2m_euclidean_fitness_ep_ = 0.01;
  int gicps2s_ransac_iter_ = 5, gicps2m_ransac_iter_ = 5;
  double gicps2s_ransac_inlier_thresh_ = 1.0, gicps2m_ransac_inlier_thresh_ = 1.0;

  OdomNode() {
    this->T = Eigen::Matrix4f::Identity();
    this->T_s2s = Eigen::Matrix4f::Identity();
    this->T_s2s_prev = Eigen::Matrix4f::Identity();
    this->imu_SE3 = Eigen::Matrix4f::Identity();

    // odom.cc:100-120
    this->gicp_s2s.setCorrespondenceRandomness(this->gicps2s_k_correspondences_);
    this->gicp_s2s.setMaxCorrespondenceDistance(this->gicps2s_max_corr_dist_);
    this->gicp_s2s.setMaximumIterations(this->gicps2s_max_iter_);
    this->gicp_s2s.setTransformationEpsilon(this->gicps2s_transformation_ep_);
    this->gicp_s2s.setEuclideanFitnessEpsilon(this->gicps2s_euclidean_fitness_ep_);
    this->gicp_s2s.setRANSACIterations(this->gicps2s_ransac_iter_);
    this->gicp_s2s.setRANSACOutlierRejectionThreshold(this->gicps2s_ransac_inlier_thresh_);

    this->gicp.setCorrespondenceRandomness(this->gicps2m_k_correspondences_);
    this->gicp.setMaxCorrespondenceDistance(this->gicps2m_max_corr_dist_);
    this->gicp.setMaximumIterations(this->gicps2m_max_iter_);
    this->gicp.setTransformationEpsilon(this->gicps2m_transformation_ep_);
    this->gicp.setEuclideanFitnessEpsilon(this->gicps2m_euclidean_fitness_ep_);
    this->gicp.setRANSACIterations(this->gicps2m_ransac_iter_);
    this->gicp.setRANSACOutlierRejectionThreshold(this->gicps2m_ransac_inlier_thresh_);

    pcl::Registration<PointType, PointType>::KdTreeReciprocalPtr temp;
    this->gicp_s2s.setSearchMethodSource(temp, true);
    this->gicp_s2s.setSearchMethodTarget(temp, true);
    this->gicp.setSearchMethodSource(temp, true);
    this->gicp.setSearchMethodTarget(temp, true);
  }

  // pcl::transformPointCloud with a float matrix (odom.cc:484,971-974); the shim offers the same on the device
  // (NanoGICP::transformCloud), this is the host form the reference node uses
  static pcl::PointCloud<PointType>::Ptr transformed(const pcl::PointCloud<PointType>& in, const Eigen::Matrix4f& M) {
    pcl::PointCloud<PointType>::Ptr out(new pcl::PointCloud<PointType>);
    for (size_t i = 0; i < in.size(); ++i) {
      const PointType& p = in.points[i];
      out->push_back(PointType(M(0, 0) * p.x + M(0, 1) * p.y + M(0, 2) * p.z + M(0, 3), M(1, 0) * p.x + M(1, 1) * p.y + M(1, 2) * p.z + M(1, 3),
                               M(2, 0) * p.x + M(2, 1) * p.y + M(2, 2) * p.z + M(2, 3), p.intensity));
    }
    return out;
  }

  void addKeyframe(const pcl::PointCloud<PointType>::Ptr& cloud_world) {
    this->keyframes.push_back(cloud_world);
    this->keyframe_cloud = pcl::PointCloud<PointType>::Ptr(new pcl::PointCloud<PointType>);
    *this->keyframe_cloud = *cloud_world;
    // odom.cc:498-500 and :1172-1174
    this->gicp_s2s.setInputSource(this->keyframe_cloud);
    this->gicp_s2s.calculateSourceCovariances();
    if (this->device_keyframes)
      this->gicp.addKeyframe(this->gicp_s2s);  // fast path: the cloud, its index and its covariances never leave the GPU
    else
      this->keyframe_normals.push_back(this->gicp_s2s.getSourceCovariances());
  }

  void initializeInputTarget() {  // odom.cc:472-507
    this->target_cloud = pcl::PointCloud<PointType>::Ptr(new pcl::PointCloud<PointType>);
    this->target_cloud = this->current_scan;
    this->gicp_s2s.setInputTarget(this->target_cloud);
    this->gicp_s2s.calculateTargetCovariances();
    addKeyframe(transformed(*this->target_cloud, this->T));
  }

  void setInputSources() {  // odom.cc:510-528
    this->gicp_s2s.setInputSource(this->current_scan);
    this->gicp.registerInputSource(this->current_scan);
    this->gicp.source_kdtree_ = this->gicp_s2s.source_kdtree_;
    this->gicp.source_covs_.clear();
  }

  void getSubmapKeyframes() {  // odom.cc:1240-1331 with every keyframe selected (the hull / kNN selection is not on the path)
    this->submap_kf_idx_curr.clear();
    for (size_t k = 0; k < this->keyframes.size(); ++k) this->submap_kf_idx_curr.push_back((int)k);
    if (this->submap_kf_idx_curr == this->submap_kf_idx_prev) {
      this->submap_hasChanged = false;
    } else {
      this->submap_hasChanged = true;
      if (!this->device_keyframes) {
        pcl::PointCloud<PointType>::Ptr submap_cloud_(new pcl::PointCloud<PointType>);
        this->submap_normals.clear();
        for (auto k : this->submap_kf_idx_curr) {
          for (const auto& p : this->keyframes[k]->points) submap_cloud_->push_back(p);  // *submap_cloud_ += *keyframes[k].second
          this->submap_normals.insert(std::end(this->submap_normals), std::begin(this->keyframe_normals[k]), std::end(this->keyframe_normals[k]));
        }
        this->submap_cloud = submap_cloud_;
      }
      this->submap_kf_idx_prev = this->submap_kf_idx_curr;
    }
  }

  void getNextPose() {  // odom.cc:795-845
    pcl::PointCloud<PointType>::Ptr aligned(new pcl::PointCloud<PointType>);

    if (this->imu_use_) {
      this->gicp_s2s.align(*aligned, this->imu_SE3);
    } else {
      this->gicp_s2s.align(*aligned);
    }

    Eigen::Matrix4f T_S2S = this->gicp_s2s.getFinalTransformation();
    print("s2s", T_S2S, this->gicp_s2s.getNrIterations(), this->gicp_s2s.hasConverged());
    this->T_s2s = this->T_s2s_prev * T_S2S;  // propagateS2S, odom.cc:905

    this->gicp.source_covs_ = this->gicp_s2s.source_covs_;
    this->gicp_s2s.swapSourceAndTarget();

    this->getSubmapKeyframes();

    if (this->submap_hasChanged) {
      if (this->device_keyframes) {
        this->gicp.setSubmapKeyframes(this->submap_kf_idx_curr);  // fast path for the two calls below
      } else {
        this->gicp.setInputTarget(this->submap_cloud);
        this->gicp.setTargetCovariances(this->submap_normals);
      }
    }

    this->gicp.align(*aligned, this->T_s2s);
    this->T = this->gicp.getFinalTransformation();
    this->T_s2s_prev = this->T;
    print("s2m", this->T, this->gicp.getNrIterations(), this->gicp.hasConverged());
    std::printf("aligned %zu %.9g %.9g %.9g %.1f\n", aligned->size(), aligned->points[7].x, aligned->points[7].y, aligned->points[7].z, aligned->points[7].data[3]);
  }

  static void print(const char* tag, const Eigen::Matrix4f& M, int iters, bool conv) {
    std::printf("%s", tag);
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) std::printf(" %.9g", M(r, c));
    std::printf(" %d %d\n", iters, conv ? 1 : 0);
  }
};

}  // namespace dlo

static pcl::PointCloud<PointType>::Ptr load(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<float> raw(bytes / 4);
  if (std::fread(raw.data(), 4, raw.size(), f) != raw.size()) std::exit(2);
  std::fclose(f);
  pcl::PointCloud<PointType>::Ptr c(new pcl::PointCloud<PointType>);
  for (size_t i = 0; i + 2 < raw.size(); i += 3) c->push_back(PointType(raw[i], raw[i + 1], raw[i + 2]));
  return c;
}

int main(int argc, char** argv) {
  int a = 1;
  bool device_keyframes = false;
  bool debug_print = false;
  if (a < argc && !std::strcmp(argv[a], "--device-keyframes")) device_keyframes = true, ++a;
  if (a < argc && !std::strcmp(argv[a], "--debug-print")) debug_print = true, ++a;  // setDebugPrint(true) on the scan-to-submap instance
  if (argc - a < 3) return 2;
  const int n = std::atoi(argv[a++]);
  std::vector<pcl::PointCloud<PointType>::Ptr> scans;
  for (int i = 0; i < n && a < argc; ++i) scans.push_back(load(argv[a++]));

  dlo::OdomNode node;
  if (!node.gicp_s2s.valid() || !node.gicp.valid()) return 3;
  node.device_keyframes = device_keyframes;
  if (debug_print) node.gicp.setDebugPrint(true);
  node.current_scan = scans[0];
  node.initializeInputTarget();
  if (!device_keyframes) {
    const auto& kn = node.keyframe_normals[0];
    std::printf("covs %zu %.12g %.12g %.12g\n", kn.size(), kn[0](0, 0), kn[0](1, 2), kn[kn.size() - 1](2, 2));
  } else {
    const auto kn = node.gicp_s2s.getSourceCovariances();
    std::printf("covs %zu %.12g %.12g %.12g\n", kn.size(), kn[0](0, 0), kn[0](1, 2), kn[kn.size() - 1](2, 2));
  }
  for (size_t i = 1; i < scans.size(); ++i) {
    node.current_scan = scans[i];
    node.setInputSources();
    node.getNextPose();
    if (i % 2 == 0) {  // a new keyframe every second scan (the reference decides by distance / rotation, odom.cc:1097-1160)
      node.current_scan_t = dlo::OdomNode::transformed(*node.current_scan, node.T);  // transformCurrentScan, odom.cc:971-974
      node.addKeyframe(node.current_scan_t);
    }
  }
  return 0;
}
