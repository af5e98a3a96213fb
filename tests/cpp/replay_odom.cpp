// Replays the exact NanoGICP call sequence of DLO's odometry node (/root/reference/src/dlo/odom.cc; line
// numbers in the comments) through the header-only shim, on clouds read from raw float32 files, and prints the
// resulting transforms.  The pytest driver compares them with the CPU oracle run through the same sequence.
//   usage: replay_odom <n_scans> <scan0.bin> <scan1.bin> ...   (each file: N x 3 float32)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "nano_gicp/nano_gicp.hpp"

using PointType = pcl::PointXYZI;  // include/dlo/dlo.h:50
using Cloud = pcl::PointCloud<PointType>;
using GICP = nano_gicp::NanoGICP<PointType, PointType>;

static Cloud::Ptr load(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<float> raw(bytes / 4);
  if (std::fread(raw.data(), 4, raw.size(), f) != raw.size()) std::exit(2);
  std::fclose(f);
  auto c = std::make_shared<Cloud>();
  for (size_t i = 0; i + 2 < raw.size(); i += 3) c->push_back(PointType(raw[i], raw[i + 1], raw[i + 2]));
  return c;
}

static void print(const char* tag, const GICP::Matrix4& T, int iters, bool conv) {
  std::printf("%s", tag);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::printf(" %.9g", T(r, c));
  std::printf(" %d %d\n", iters, conv ? 1 : 0);
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  int n = std::atoi(argv[1]);
  std::vector<Cloud::Ptr> scans;
  for (int i = 0; i < n; ++i) scans.push_back(load(argv[2 + i]));

  GICP gicp_s2s, gicp;  // include/dlo/odom.h:119-120
  if (!gicp_s2s.valid() || !gicp.valid()) return 3;
  gicp_s2s.setCorrespondenceRandomness(10);  // odom.cc:100-106 with cfg/params.yaml:54-62
  gicp_s2s.setMaxCorrespondenceDistance(1.0);
  gicp_s2s.setMaximumIterations(32);
  gicp_s2s.setTransformationEpsilon(0.01);
  gicp_s2s.setEuclideanFitnessEpsilon(0.01);
  gicp_s2s.setRANSACIterations(5);
  gicp_s2s.setRANSACOutlierRejectionThreshold(1.0);
  gicp.setCorrespondenceRandomness(20);      // odom.cc:108-114 with cfg/params.yaml:63-71
  gicp.setMaxCorrespondenceDistance(0.5);
  gicp.setMaximumIterations(32);
  gicp.setTransformationEpsilon(0.01);
  gicp.setEuclideanFitnessEpsilon(0.01);
  gicp.setRANSACIterations(5);
  gicp.setRANSACOutlierRejectionThreshold(1.0);
  GICP::KdTreeReciprocalPtr temp;            // odom.cc:116-120
  gicp_s2s.setSearchMethodSource(temp, true);
  gicp_s2s.setSearchMethodTarget(temp, true);
  gicp.setSearchMethodSource(temp, true);
  gicp.setSearchMethodTarget(temp, true);

  // first scan: initializeInputTarget()  odom.cc:472-507
  gicp_s2s.setInputTarget(scans[0]);
  gicp_s2s.calculateTargetCovariances();
  gicp_s2s.setInputSource(scans[0]);
  gicp_s2s.calculateSourceCovariances();
  GICP::CovVector keyframe_normals = gicp_s2s.getSourceCovariances();
  std::printf("covs %zu %.12g %.12g %.12g\n", keyframe_normals.size(), keyframe_normals[0](0, 0), keyframe_normals[0](1, 2),
              keyframe_normals[keyframe_normals.size() - 1](2, 2));
  Cloud::Ptr submap_cloud = scans[0];
  GICP::Matrix4 T_prev = GICP::Matrix4::Identity();

  for (int i = 1; i < n; ++i) {
    gicp_s2s.setInputSource(scans[i]);             // setInputSources()  odom.cc:519
    gicp.registerInputSource(scans[i]);            // odom.cc:522
    gicp.source_kdtree_ = gicp_s2s.source_kdtree_; // odom.cc:525
    gicp.source_covs_.clear();                     // odom.cc:526
    Cloud::Ptr aligned(new Cloud);
    gicp_s2s.align(*aligned);                      // getNextPose()  odom.cc:805
    GICP::Matrix4 T_S2S = gicp_s2s.getFinalTransformation();
    print("s2s", T_S2S, gicp_s2s.getNrIterations(), gicp_s2s.hasConverged());
    gicp.source_covs_ = gicp_s2s.source_covs_;     // odom.cc:815
    gicp_s2s.swapSourceAndTarget();                // odom.cc:818
    if (i == 1) {                                  // submap changed  odom.cc:827-834
      gicp.setInputTarget(submap_cloud);
      gicp.setTargetCovariances(keyframe_normals);
    }
    GICP::Matrix4 guess = T_prev * T_S2S;
    gicp.align(*aligned, guess);                   // odom.cc:837
    T_prev = gicp.getFinalTransformation();        // odom.cc:840
    print("s2m", T_prev, gicp.getNrIterations(), gicp.hasConverged());
    std::printf("aligned %zu %.9g %.9g %.9g %.1f\n", aligned->size(), aligned->points[7].x, aligned->points[7].y, aligned->points[7].z, aligned->points[7].data[3]);
  }
  return 0;
}
