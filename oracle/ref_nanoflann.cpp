// =============================================================================
// ref_nanoflann.cpp — thin C-ABI adaptor around the REAL reference kd-tree.
//
// *** TEST INFRASTRUCTURE ONLY ***  Built only in the authoring container,
// where /root/reference exists:   make -C oracle ref
// The header it includes stays where it lies under /root/reference (it is
// never copied into this repo); the resulting oracle/_ref/libref_nanoflann.so
// is git-ignored and travels to the GPU box as a prebuilt binary.
//
// The instantiation mirrors /root/reference/include/nano_gicp/nanoflann.hpp:
//   :100-102  KDTreeSingleIndexAdaptor<SO3_Adaptor<float, Adaptor>, Adaptor, 3, int>
//   :113-117  leaf_max_size = 100
//   :141-152  nearestKSearch -> KNNResultSet<float,int> + findNeighbors(default SearchParams)
//   :185-191  kdtree_get_pt returns x / y / z
// The PCL wrapper itself (nanoflann.hpp) needs PCL + boost and cannot be
// compiled here; it is a pass-through to exactly the calls made below.
// =============================================================================
#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

#include <nano_gicp/impl/nanoflann_impl.hpp>  // from -I/root/reference/include

namespace {

struct XyzAdaptor {
  const float* base = nullptr;
  size_t count = 0;
  size_t stride = 0;  // floats
  inline size_t kdtree_get_point_count() const { return count; }
  inline float kdtree_get_pt(const size_t idx, int dim) const { return dim < 3 ? base[idx * stride + dim] : 0.0f; }
  template <class BBOX>
  bool kdtree_get_bbox(BBOX&) const {
    return false;
  }
};

using RefTree = nanoflann::KDTreeSingleIndexAdaptor<nanoflann::SO3_Adaptor<float, XyzAdaptor>, XyzAdaptor, 3, int>;

}  // namespace

extern "C" {

struct ref_tree {
  std::vector<float> pts;
  XyzAdaptor adaptor;
  RefTree* tree = nullptr;
};

ref_tree* ref_tree_build(const float* xyz, size_t n, size_t stride_floats) {
  ref_tree* t = new ref_tree;
  t->pts.resize(n * 4);
  for (size_t i = 0; i < n; ++i) {
    for (int d = 0; d < 3; ++d) t->pts[i * 4 + d] = xyz[i * stride_floats + d];
    t->pts[i * 4 + 3] = 1.0f;
  }
  t->adaptor.base = t->pts.data();
  t->adaptor.count = n;
  t->adaptor.stride = 4;
  t->tree = new RefTree(3, t->adaptor, nanoflann::KDTreeSingleIndexAdaptorParams(100));
  t->tree->buildIndex();
  return t;
}

void ref_tree_free(ref_tree* t) {
  if (!t) return;
  delete t->tree;
  delete t;
}

int ref_tree_knn(const ref_tree* t, const float* queries, size_t nq, size_t qstride_floats, int k, int* idx, float* d2, int threads) {
  if (!t || k <= 0) return -1;
  if (threads <= 0) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(guided, 8)
  for (long i = 0; i < (long)nq; ++i) {
    nanoflann::KNNResultSet<float, int> rs(k);
    rs.init(idx + (size_t)i * k, d2 + (size_t)i * k);
    t->tree->findNeighbors(rs, queries + (size_t)i * qstride_floats, nanoflann::SearchParams());
    for (int j = (int)rs.size(); j < k; ++j) {
      idx[(size_t)i * k + j] = -1;
      d2[(size_t)i * k + j] = std::numeric_limits<float>::infinity();
    }
  }
  return 0;
}

}  // extern "C"
