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
// Built with -ffp-contract=off like the rest (no FMA).  Every kernel is written out below (round 2 called hipCUB for the sort), and
// the counts that size each step stay on the device: one host synchronisation per call, behind the last kernel.
#include <hip/hip_runtime.h>

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

// What the filter's kernels hand each other WITHOUT a host visit (the host reads it once, behind the last kernel)
struct FilterState {
  int n_surv;    // points that survive removeNaN / CropBox
  int n_vox;     // occupied voxels
  int overflow;  // div.x * div.y * div.z does not fit an int: PCL warns and returns its input
  int bits;      // significant bits of a voxel index
  Lattice L;
};

// one block: survivors' bounding box (per-block partials) -> the voxel lattice (pcl/filters/impl/voxel_grid.hpp: min_b / max_b / div_b)
__global__ void __launch_bounds__(256) k_lattice(const int* __restrict__ n_surv_ptr, const float* __restrict__ bbox_part, int nparts, float leaf, FilterState* __restrict__ st) {
  __shared__ float lds[4][6];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int b = threadIdx.x; b < nparts; b += blockDim.x)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = fminf(mn[d], bbox_part[b * 8 + d]);
      mx[d] = fmaxf(mx[d], bbox_part[b * 8 + 3 + d]);
    }
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lds[wave][d] = mn[d];
      lds[wave][3 + d] = mx[d];
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    FilterState r;
    r.n_surv = *n_surv_ptr;
    r.n_vox = 0;
    r.overflow = 0;
    r.bits = 1;
    r.L.inv_leaf = leaf > 0.f ? 1.0f / leaf : 0.f;
    long long cells = 1;
    for (int d = 0; d < 3; ++d) {
      float lo = lds[0][d], hi = lds[0][3 + d];
      for (int w = 1; w < 4; ++w) {
        lo = fminf(lo, lds[w][d]);
        hi = fmaxf(hi, lds[w][3 + d]);
      }
      if (r.n_surv <= 0 || !(leaf > 0.f)) lo = hi = 0.f;
      r.L.min_b[d] = (int)floorf(lo * r.L.inv_leaf);
      const int max_b = (int)floorf(hi * r.L.inv_leaf);
      r.L.div[d] = max_b - r.L.min_b[d] + 1;
      if (!r.overflow) {
        cells *= (long long)r.L.div[d];
        if (cells > 0x7fffffffll) r.overflow = 1;
      }
    }
    if (!r.overflow)
      while ((1ll << r.bits) < cells && r.bits < 32) ++r.bits;
    *st = r;
  }
}

__global__ void __launch_bounds__(256) k_voxel_keys(const float4* __restrict__ pts, const FilterState* __restrict__ st, unsigned int* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st->n_surv || st->overflow) return;
  const Lattice L = st->L;
  const float4 p = pts[i];
  const int ijk0 = (int)floorf(p.x * L.inv_leaf) - L.min_b[0];
  const int ijk1 = (int)floorf(p.y * L.inv_leaf) - L.min_b[1];
  const int ijk2 = (int)floorf(p.z * L.inv_leaf) - L.min_b[2];
  keys[i] = (unsigned int)(ijk0 + ijk1 * L.div[0] + ijk2 * L.div[0] * L.div[1]);
  vals[i] = i;
}

// ---- stable LSD radix sort of (voxel index, point index) pairs, 11 bits per pass, written out here (round 2 called hipCUB) ----
// A WAVE owns a tile of kRadixTile consecutive elements and walks it in order, 64 at a time.  Pass structure: (1) per-wave digit
// histogram -> hist[digit][wave]; (2) exclusive scan over hist in that (digit-major) order = where every wave's elements of every
// digit go; (3) the same walk again: inside a step a lane's rank among the lanes with its digit comes from ballots (one per digit bit:
// no LDS atomics, no per-lane loops), the running position of every digit sits in LDS.  Order inside a digit = wave order, then step
// order, then lane order = input order: stable.  A pass whose bits are all above the highest bit in use copies.
constexpr int kRadixBits = 11, kRadixBins = 1 << kRadixBits, kRadixTile = 1024;

__global__ void __launch_bounds__(256) k_radix_hist(const unsigned int* __restrict__ keys, const FilterState* __restrict__ st, int shift, int nwaves, int* __restrict__ hist) {
  __shared__ int h[4][kRadixBins];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gw = blockIdx.x * 4 + wave;
  if (shift >= st->bits || st->overflow) return;  // (block-uniform)
  for (int b = lane; b < kRadixBins; b += 64) h[wave][b] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int n = st->n_surv, base = gw * kRadixTile;
  if (gw < nwaves)
    for (int o = lane; o < kRadixTile; o += 64) {
      const int i = base + o;
      if (i < n) atomicAdd(&h[wave][(keys[i] >> shift) & (kRadixBins - 1)], 1);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (gw < nwaves)
    for (int b = lane; b < kRadixBins; b += 64) hist[(size_t)b * nwaves + gw] = h[wave][b];
}

__global__ void __launch_bounds__(256) k_radix_scatter(const unsigned int* __restrict__ keys, const int* __restrict__ vals, const FilterState* __restrict__ st, int shift, int nwaves,
                                                        const int* __restrict__ offs, unsigned int* __restrict__ keys_out, int* __restrict__ vals_out) {
  __shared__ int run[4][kRadixBins];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gw = blockIdx.x * 4 + wave;
  if (gw >= nwaves || st->overflow) return;  // (wave-uniform; no block-level synchronisation below)
  const int n = st->n_surv, base = gw * kRadixTile;
  if (shift >= st->bits) {  // nothing to sort by: copy
    for (int o = lane; o < kRadixTile; o += 64) {
      const int i = base + o;
      if (i < n) {
        keys_out[i] = keys[i];
        vals_out[i] = vals[i];
      }
    }
    return;
  }
  for (int b = lane; b < kRadixBins; b += 64) run[wave][b] = offs[(size_t)b * nwaves + gw];
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int o = 0; o < kRadixTile; o += 64) {  // in order: stability
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int i = base + o + lane;
    const bool valid = i < n;
    const unsigned int key = valid ? keys[i] : 0u;
    const int val = valid ? vals[i] : 0;
    const int d = (int)((key >> shift) & (kRadixBins - 1));
    unsigned long long m = __ballot(valid);  // the lanes that share this lane's digit
#pragma unroll
    for (int bit = 0; bit < kRadixBits; ++bit) {
      const bool one = (d >> bit) & 1;
      const unsigned long long bb = __ballot(valid && one);
      m &= one ? bb : ~bb;
    }
    if (valid) {
      const int pos = run[wave][d] + __popcll(m & lt);
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (valid && (m & lt) == 0) run[wave][d] += __popcll(m);  // the digit's lowest lane moves its running position
    if (__ballot(valid) == 0) break;
  }
}

// head[j] = 1 where a new voxel starts in the sorted key sequence (0 behind the last survivor: the scan runs over the input size)
__global__ void __launch_bounds__(256) k_voxel_heads(const unsigned int* __restrict__ keys, const FilterState* __restrict__ st, int n_cap, int* __restrict__ head) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_cap) return;
  head[j] = (j < st->n_surv && !st->overflow && (j == 0 || keys[j] != keys[j - 1])) ? 1 : 0;
}

// seg_start[v] = first sorted position of voxel v (v = exclusive prefix of head at a head position); seg_start[n_vox] = n; the number of voxels
__global__ void __launch_bounds__(256) k_voxel_starts(const int* __restrict__ head, const int* __restrict__ vox_of, FilterState* __restrict__ st, int* __restrict__ seg_start) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = st->n_surv;
  if (j >= n || st->overflow) return;
  if (head[j]) seg_start[vox_of[j]] = j;
  if (j == n - 1) {
    seg_start[vox_of[j] + head[j]] = n;  // vox_of is the EXCLUSIVE prefix: the last voxel's id is vox_of[n-1] + head[n-1] - 1
    st->n_vox = vox_of[j] + head[j];
  }
}

// one thread per voxel: centroid of all four fields, float sums in input order, divided by the count
__global__ void __launch_bounds__(256) k_voxel_centroids(const float4* __restrict__ pts, const int* __restrict__ order, const int* __restrict__ seg_start, const FilterState* __restrict__ st,
                                                          float4* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= st->n_vox || st->overflow) return;
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

// Everything up to the last kernel is enqueued without a host visit: the survivors' count, their bounding box, the lattice and the
// voxel count stay on the device (FilterState); launches are sized for the input (an upper bound) and return early beyond the
// counts.  ONE synchronisation at the end reads the state (16 bytes).
static int filter_cloud_impl(hipStream_t s, FilterWorkspace* ws, const float4* in_dev, int n, int remove_nan, float crop_half, float leaf, const float4** out_dev,
                             int* n_out, char* err, size_t errlen, bool force_all_passes) {
  int passes_run = 3;
  *out_dev = in_dev;
  *n_out = n;
  if (n <= 0) return 0;
  const bool crop = crop_half > 0.f, voxel = leaf > 0.f;
  const int blocks = (n + 255) / 256, cblocks = std::min(512, (n + 1023) / 1024);
  const int nwaves = (n + kRadixTile - 1) / kRadixTile, rblocks = (nwaves + 3) / 4;
  const size_t hist_entries = (size_t)kRadixBins * nwaves;
  // (every buffer is sized BEFORE the first launch: growing one frees it, and freeing waits for the device)
  if (ensure(ws, 7, (std::max(hist_entries, (size_t)n) / kScanTile + 2) * sizeof(int), err, errlen)) return -1;
  if (ensure(ws, 0, (size_t)n * sizeof(unsigned int) * 2, err, errlen) || ensure(ws, 1, (size_t)(n + 1 + kCellPad) * sizeof(int) * 2, err, errlen) ||
      ensure(ws, 2, (size_t)n * sizeof(float4), err, errlen) || ensure(ws, 3, (size_t)cblocks * 8 * sizeof(float) + sizeof(FilterState) + 64, err, errlen))
    return -1;
  int* keep = reinterpret_cast<int*>(ws->buf[0]);
  int* offs = reinterpret_cast<int*>(ws->buf[1]);
  float4* comp = reinterpret_cast<float4*>(ws->buf[2]);
  float* bbox = reinterpret_cast<float*>(ws->buf[3]);
  FilterState* st = reinterpret_cast<FilterState*>(reinterpret_cast<char*>(ws->buf[3]) + (((size_t)cblocks * 8 * sizeof(float) + 63) / 64) * 64);
  // ---- removeNaN + CropBox: flags, exclusive scan, order-preserving compaction (+ the survivors' bounding box) ----
  hipLaunchKernelGGL(k_filter_flags, dim3(blocks), dim3(256), 0, s, in_dev, n, (remove_nan || voxel) ? 1 : 0, crop ? crop_half : 0.f, keep);
  if (exclusive_scan(s, ws, keep, n, offs, err, errlen)) return -1;
  hipLaunchKernelGGL(k_filter_compact, dim3(cblocks), dim3(256), 0, s, in_dev, (const int*)keep, (const int*)offs, n, comp, bbox);
  hipLaunchKernelGGL(k_lattice, dim3(1), dim3(256), 0, s, (const int*)(offs + n), (const float*)bbox, cblocks, voxel ? leaf : 0.f, st);
  float4* out = nullptr;
  if (voxel) {
    // ---- VoxelGrid: voxel index per survivor, stable sort, segment heads, one centroid per voxel ----
    if (ensure(ws, 4, (size_t)n * sizeof(int) * 2, err, errlen) || ensure(ws, 5, (hist_entries + 1 + kCellPad) * sizeof(int) * 2, err, errlen) ||
        ensure(ws, 6, (size_t)n * sizeof(float4), err, errlen))
      return -1;
    unsigned int* keys_a = reinterpret_cast<unsigned int*>(ws->buf[0]);  // (the flags are done with it)
    unsigned int* keys_b = keys_a + n;
    int* vals_a = reinterpret_cast<int*>(ws->buf[4]);
    int* vals_b = vals_a + n;
    int* hist = reinterpret_cast<int*>(ws->buf[5]);
    int* hoffs = hist + hist_entries;  // exclusive prefix (hist_entries + 1 + pad)
    hipLaunchKernelGGL(k_voxel_keys, dim3(blocks), dim3(256), 0, s, (const float4*)comp, (const FilterState*)st, keys_a, vals_a);
    // 3 x 11 bits cover the 31 a voxel index may have, but consecutive scans / map updates need the same number of bits: the passes
    // the PREVIOUS call of this workspace needed are enqueued, and the state read back at the end says whether that was enough (if
    // not - first call, or the scene grew - the sort is simply run again with all three)
    const int passes = (ws->last_bits > 0 && !force_all_passes) ? std::min(3, (ws->last_bits + kRadixBits - 1) / kRadixBits) : 3;
    passes_run = passes;
    for (int pass = 0; pass < passes; ++pass) {
      hipLaunchKernelGGL(k_radix_hist, dim3(rblocks), dim3(256), 0, s, (const unsigned int*)keys_a, (const FilterState*)st, pass * kRadixBits, nwaves, hist);
      if (exclusive_scan(s, ws, hist, (int)hist_entries, hoffs, err, errlen)) return -1;
      hipLaunchKernelGGL(k_radix_scatter, dim3(rblocks), dim3(256), 0, s, (const unsigned int*)keys_a, (const int*)vals_a, (const FilterState*)st, pass * kRadixBits, nwaves,
                         (const int*)hoffs, keys_b, vals_b);
      std::swap(keys_a, keys_b);
      std::swap(vals_a, vals_b);
    }
    int* head = reinterpret_cast<int*>(keys_b);                       // (the other half of the ping-pong is free now)
    int* vox_of = reinterpret_cast<int*>(ws->buf[1]);                 // exclusive prefix of the heads (n + 1 + pad); the compaction offsets are done
    int* seg_start = vox_of + (n + 1 + kCellPad);                     // n + 1 entries at most
    hipLaunchKernelGGL(k_voxel_heads, dim3(blocks), dim3(256), 0, s, (const unsigned int*)keys_a, (const FilterState*)st, n, head);
    if (exclusive_scan(s, ws, head, n, vox_of, err, errlen)) return -1;
    hipLaunchKernelGGL(k_voxel_starts, dim3(blocks), dim3(256), 0, s, (const int*)head, (const int*)vox_of, st, seg_start);
    out = reinterpret_cast<float4*>(ws->buf[6]);
    hipLaunchKernelGGL(k_voxel_centroids, dim3(blocks), dim3(256), 0, s, (const float4*)comp, (const int*)vals_a, (const int*)seg_start, (const FilterState*)st, out);
  }
  FilterState hs;
  FLT_TRY(hipMemcpyAsync(&hs, st, sizeof(hs), hipMemcpyDeviceToHost, s));
  FLT_TRY(hipStreamSynchronize(s));
  FLT_TRY(hipGetLastError());
  *out_dev = comp;
  *n_out = hs.n_surv;
  if (!voxel || hs.n_surv == 0) return 0;
  if (hs.overflow) {
    std::fprintf(stderr, "[VoxelGrid] Leaf size is too small for the input dataset. Integer indices would overflow.\n");  // PCL: output = input
    return 0;
  }
  ws->last_bits = hs.bits;
  if (hs.bits > passes_run * kRadixBits) return filter_cloud_impl(s, ws, in_dev, n, remove_nan, crop_half, leaf, out_dev, n_out, err, errlen, true);  // (sorted on too few bits)
  *out_dev = out;
  *n_out = hs.n_vox;
  return 0;
}

extern "C" int ngk_filter_cloud(hipStream_t s, FilterWorkspace* ws, const float4* in_dev, int n, int remove_nan, float crop_half, float leaf, const float4** out_dev,
                                int* n_out, char* err, size_t errlen) {
  return filter_cloud_impl(s, ws, in_dev, n, remove_nan, crop_half, leaf, out_dev, n_out, err, errlen, false);
}

extern "C" void ngk_filter_free(FilterWorkspace* ws) {
  for (int i = 0; i < 8; ++i) {
    if (ws->buf[i]) (void)hipFree(ws->buf[i]);
    ws->buf[i] = nullptr;
    ws->cap[i] = 0;
  }
}
