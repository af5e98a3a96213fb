// Scan preprocessing and map voxel filtering on the GPU:
//   pcl::removeNaNFromPointCloud + pcl::CropBox (negative) + pcl::VoxelGrid   /root/reference/src/dlo/odom.cc:443-465, 122-127
//   pcl::VoxelGrid over the accumulated map                                   /root/reference/src/dlo/map.cc:100-131
// PCL is a third-party dependency that is NOT under /root/reference (PCL >= 1.10, unpinned: README.md:27) and is not
// installed here: the filters restate pcl/filters/impl/{crop_box,voxel_grid}.hpp and pcl/filters/filter.hpp FROM MEMORY
// (parity unpinned; the oracle restates the same rules on the CPU, oracle/ngicp_oracle.cpp "filters"):
//   removeNaN   keeps the points whose x, y, z are all finite, in order;
//   CropBox     (no box pose; negative = true) drops the points with min <= p <= max in all three coordinates, keeps the
//               rest in order;
//   VoxelGrid   leaf L (cubic), inverse leaf = 1/L in float; lattice anchored at the origin: ijk = floor(p * inv_leaf);
//               min_b / max_b from the bounding box of the input, div_b = max_b - min_b + 1; voxel index
//               (i - min_b.x) + (j - min_b.y) * div.x + (k - min_b.z) * div.x * div.y; one output point per occupied voxel,
//               in ascending voxel index, = the centroid of ALL fields (x, y, z, intensity: downsample_all_data), float sums
//               divided by the count; if div.x * div.y * div.z overflows int32 PCL warns and returns the input unfiltered.
//               PCL adds the points of a voxel in the order std::sort leaves them (unspecified among equal indices); here
//               they are added in input order (a stable radix sort), so sums may differ from PCL's in the last bits.
// Built with -ffp-contract=off like the rest (no FMA).  The radix sort is hipCUB's (stable, deterministic); everything else
// is written out below.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "ngicp_filters.h"

namespace {  // this translation unit's own (internal-linkage) copy of the grid kernels: only the three-kernel exclusive scan is used
#include "ngicp_grid.h"
}  // namespace

using namespace ngk;

namespace {

#define FLT_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      std::snprintf(err, errlen, "HIP error %d (%s) in `%s`", (int)_e, hipGetErrorString(_e), #expr); \
      return -1;                                                                              \
    }                                                                                         \
  } while (0)

int ensure(FilterWorkspace* ws, int slot, size_t bytes, char* err, size_t errlen) {
  if (bytes <= ws->cap[slot]) return 0;
  if (ws->buf[slot]) FLT_TRY(hipFree(ws->buf[slot]));
  ws->buf[slot] = nullptr;
  ws->cap[slot] = 0;
  const size_t want = bytes + bytes / 4 + 256;
  FLT_TRY(hipMalloc(&ws->buf[slot], want));
  ws->cap[slot] = want;
  return 0;
}

// keep[i] = 1 when point i survives removeNaN / CropBox
__global__ void __launch_bounds__(256) k_filter_flags(const float4* __restrict__ pts, int n, int drop_nonfinite, float crop, int* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  bool k = true;
  if (drop_nonfinite && !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) k = false;
  // pcl::CropBox, negative: a point INSIDE [min, max] (inclusive) is removed
  if (k && crop > 0.f && !(p.x < -crop || p.y < -crop || p.z < -crop || p.x > crop || p.y > crop || p.z > crop)) k = false;
  keep[i] = k ? 1 : 0;
}

// order-preserving compaction + per-block bounding box of the survivors
__global__ void __launch_bounds__(256) k_filter_compact(const float4* __restrict__ pts, const int* __restrict__ keep, const int* __restrict__ offs, int n,
                                                         float4* __restrict__ out, float* __restrict__ bbox_part) {
  __shared__ float lds[4][6];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!keep[i]) continue;
    const float4 p = pts[i];
    out[offs[i]] = p;
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lds[wave][d] = mn[d];
      lds[wave][3 + d] = mx[d];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int d = threadIdx.x;
    float v = lds[0][d];
    for (int w = 1; w < 4; ++w) v = d < 3 ? fminf(v, lds[w][d]) : fmaxf(v, lds[w][d]);
    bbox_part[blockIdx.x * 8 + d] = v;
  }
}

struct Lattice {
  float inv_leaf;
  int min_b[3];
  int div[3];
};

__global__ void __launch_bounds__(256) k_voxel_keys(const float4* __restrict__ pts, int n, Lattice L, unsigned int* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const int ijk0 = (int)floorf(p.x * L.inv_leaf) - L.min_b[0];
  const int ijk1 = (int)floorf(p.y * L.inv_leaf) - L.min_b[1];
  const int ijk2 = (int)floorf(p.z * L.inv_leaf) - L.min_b[2];
  keys[i] = (unsigned int)(ijk0 + ijk1 * L.div[0] + ijk2 * L.div[0] * L.div[1]);
  vals[i] = i;
}

// head[j] = 1 where a new voxel starts in the sorted key sequence
__global__ void __launch_bounds__(256) k_voxel_heads(const unsigned int* __restrict__ keys, int n, int* __restrict__ head) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  head[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1 : 0;
}

// seg_start[v] = first sorted position of voxel v (v = exclusive prefix of head at a head position); seg_start[n_vox] = n
__global__ void __launch_bounds__(256) k_voxel_starts(const int* __restrict__ head, const int* __restrict__ vox_of, int n, int* __restrict__ seg_start) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (head[j]) seg_start[vox_of[j]] = j;
  if (j == n - 1) seg_start[vox_of[j] + head[j]] = n;  // vox_of is the EXCLUSIVE prefix: the last voxel's id is vox_of[n-1] + head[n-1] - 1
}

// one thread per voxel: centroid of all four fields, float sums in input order, divided by the count
__global__ void __launch_bounds__(256) k_voxel_centroids(const float4* __restrict__ pts, const int* __restrict__ order, const int* __restrict__ seg_start, int n_vox,
                                                          float4* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vox) return;
  const int s = seg_start[v], e = seg_start[v + 1];
  float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
  for (int j = s; j < e; ++j) {
    const float4 p = pts[order[j]];
    sx += p.x; sy += p.y; sz += p.z; si += p.w;
  }
  const float cnt = (float)(e - s);
  out[v] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
}

int exclusive_scan(hipStream_t s, FilterWorkspace* ws, const int* in, int n, int* out, char* err, size_t errlen) {  // out has n + 1 + kCellPad entries
  const int ntiles = (n + kScanTile - 1) / kScanTile;
  if (ensure(ws, 7, (size_t)ntiles * sizeof(int), err, errlen)) return -1;
  int* tile_sums = reinterpret_cast<int*>(ws->buf[7]);
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(kScanBlock), 0, s, in, n, tile_sums, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kScanBlock), 0, s, tile_sums, ntiles, (const unsigned long long*)nullptr, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(kScanBlock), 0, s, in, n, (const int*)tile_sums, out);
  return 0;
}

}  // namespace

extern "C" int ngk_filter_cloud(hipStream_t s, FilterWorkspace* ws, const float4* in_dev, int n, int remove_nan, float crop_half, float leaf, const float4** out_dev,
                                int* n_out, char* err, size_t errlen) {
  *out_dev = in_dev;
  *n_out = n;
  if (n <= 0) return 0;
  const bool crop = crop_half > 0.f, voxel = leaf > 0.f;
  const float4* cur = in_dev;
  int cur_n = n;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  {
    // ---- removeNaN + CropBox: flags, exclusive scan, order-preserving compaction (+ the survivors' bounding box) ----
    const int blocks = (n + 255) / 256, cblocks = std::min(512, (n + 1023) / 1024);
    if (ensure(ws, 0, (size_t)n * sizeof(int), err, errlen) || ensure(ws, 1, (size_t)(n + 1 + kCellPad) * sizeof(int), err, errlen) ||
        ensure(ws, 2, (size_t)n * sizeof(float4), err, errlen) || ensure(ws, 3, (size_t)cblocks * 8 * sizeof(float), err, errlen))
      return -1;
    int* keep = reinterpret_cast<int*>(ws->buf[0]);
    int* offs = reinterpret_cast<int*>(ws->buf[1]);
    float4* comp = reinterpret_cast<float4*>(ws->buf[2]);
    float* bbox = reinterpret_cast<float*>(ws->buf[3]);
    hipLaunchKernelGGL(k_filter_flags, dim3(blocks), dim3(256), 0, s, in_dev, n, (remove_nan || voxel) ? 1 : 0, crop ? crop_half : 0.f, keep);
    if (exclusive_scan(s, ws, keep, n, offs, err, errlen)) return -1;
    hipLaunchKernelGGL(k_filter_compact, dim3(cblocks), dim3(256), 0, s, in_dev, (const int*)keep, (const int*)offs, n, comp, bbox);
    float hb[512 * 8];
    int total = 0;
    FLT_TRY(hipMemcpyAsync(&total, offs + n, sizeof(int), hipMemcpyDeviceToHost, s));
    FLT_TRY(hipMemcpyAsync(hb, bbox, (size_t)cblocks * 8 * sizeof(float), hipMemcpyDeviceToHost, s));
    FLT_TRY(hipStreamSynchronize(s));
    cur = comp;
    cur_n = total;
    for (int d = 0; d < 3; ++d) mn[d] = 3.0e38f, mx[d] = -3.0e38f;
    for (int b = 0; b < cblocks; ++b)
      for (int d = 0; d < 3; ++d) {
        mn[d] = std::min(mn[d], hb[b * 8 + d]);
        mx[d] = std::max(mx[d], hb[b * 8 + 3 + d]);
      }
  }
  *out_dev = cur;
  *n_out = cur_n;
  if (!voxel || cur_n == 0) return 0;
  // ---- VoxelGrid ----
  Lattice L;
  L.inv_leaf = 1.0f / leaf;
  long long cells = 1;
  for (int d = 0; d < 3; ++d) {
    L.min_b[d] = (int)std::floor(mn[d] * L.inv_leaf);
    const int max_b = (int)std::floor(mx[d] * L.inv_leaf);
    L.div[d] = max_b - L.min_b[d] + 1;
    cells *= (long long)L.div[d];
    if (cells > (long long)INT_MAX) {
      std::fprintf(stderr, "[VoxelGrid] Leaf size is too small for the input dataset. Integer indices would overflow.\n");  // PCL: output = input
      return 0;
    }
  }
  int bits = 1;
  while ((1ll << bits) < cells && bits < 32) ++bits;
  const int blocks = (cur_n + 255) / 256;
  if (ensure(ws, 0, (size_t)cur_n * sizeof(unsigned int) * 2, err, errlen) || ensure(ws, 4, (size_t)cur_n * sizeof(int) * 2, err, errlen) ||
      ensure(ws, 1, (size_t)(cur_n + 1 + kCellPad) * sizeof(int) * 2, err, errlen))
    return -1;
  unsigned int* keys_in = reinterpret_cast<unsigned int*>(ws->buf[0]);
  unsigned int* keys_out = keys_in + cur_n;
  int* vals_in = reinterpret_cast<int*>(ws->buf[4]);
  int* vals_out = vals_in + cur_n;
  int* vox_of = reinterpret_cast<int*>(ws->buf[1]);                 // exclusive prefix of the heads (n + 1 + pad)
  int* seg_start = vox_of + (cur_n + 1 + kCellPad);                 // n + 1 entries at most
  hipLaunchKernelGGL(k_voxel_keys, dim3(blocks), dim3(256), 0, s, cur, cur_n, L, keys_in, vals_in);
  size_t tmp_bytes = 0;
  FLT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, cur_n, 0, bits, s));
  if (ensure(ws, 5, tmp_bytes, err, errlen)) return -1;
  FLT_TRY(hipcub::DeviceRadixSort::SortPairs(ws->buf[5], tmp_bytes, keys_in, keys_out, vals_in, vals_out, cur_n, 0, bits, s));
  // heads reuse the (now free) unsorted key array
  int* head = reinterpret_cast<int*>(keys_in);
  hipLaunchKernelGGL(k_voxel_heads, dim3(blocks), dim3(256), 0, s, (const unsigned int*)keys_out, cur_n, head);
  if (exclusive_scan(s, ws, head, cur_n, vox_of, err, errlen)) return -1;
  hipLaunchKernelGGL(k_voxel_starts, dim3(blocks), dim3(256), 0, s, (const int*)head, (const int*)vox_of, cur_n, seg_start);
  int n_vox = 0;
  FLT_TRY(hipMemcpyAsync(&n_vox, vox_of + cur_n, sizeof(int), hipMemcpyDeviceToHost, s));
  FLT_TRY(hipStreamSynchronize(s));
  if (ensure(ws, 6, (size_t)n_vox * sizeof(float4), err, errlen)) return -1;
  float4* out = reinterpret_cast<float4*>(ws->buf[6]);
  hipLaunchKernelGGL(k_voxel_centroids, dim3((n_vox + 255) / 256), dim3(256), 0, s, cur, (const int*)vals_out, (const int*)seg_start, n_vox, out);
  FLT_TRY(hipGetLastError());
  *out_dev = out;
  *n_out = n_vox;
  return 0;
}

extern "C" void ngk_filter_free(FilterWorkspace* ws) {
  for (int i = 0; i < 8; ++i) {
    if (ws->buf[i]) (void)hipFree(ws->buf[i]);
    ws->buf[i] = nullptr;
    ws->cap[i] = 0;
  }
}
