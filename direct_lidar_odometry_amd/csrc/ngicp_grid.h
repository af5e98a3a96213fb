// Voxel-grid spatial index (the GPU replacement of the reference's serial kd-tree build,
// /root/reference/include/nano_gicp/impl/nanoflann_impl.hpp:1199-1211,867-1012).
//
// Layout in HBM per indexed cloud:
//   sorted[n]      float4  {x, y, z, bitcast(original index)}  points in cell-major order
//   cell_start[c]  int     first sorted position of linear cell c (ncells + 1 entries)
// Linear cell id = (cz * ny + cy) * nx + cx (x fastest), so the 3 x-neighbours of a 27-cell
// probe are ONE contiguous run of `sorted`: a ring-1 probe is 9 contiguous runs.
// Points outside the grid are clamped into the border cells; the search bounds below stay
// valid because a clamped point is never closer than its border cell's inner face.
//
// The build is a deterministic counting sort: histogram (int atomics) -> exclusive scan ->
// scatter (int atomics; order inside a cell arbitrary) -> rank by (x, original index) inside each
// cell.  The final order is therefore a pure function of the input (bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ngk {

struct Grid {
  float ox, oy, oz;  // origin (min corner)
  float h, inv_h;    // voxel edge
  int nx, ny, nz;
  int ncells;
  float slack;       // conservative shrink of face distances (float cell-assignment rounding)
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void cell_coords(const Grid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
  cx = clampi((int)floorf((x - g.ox) * g.inv_h), 0, g.nx - 1);
  cy = clampi((int)floorf((y - g.oy) * g.inv_h), 0, g.ny - 1);
  cz = clampi((int)floorf((z - g.oz) * g.inv_h), 0, g.nz - 1);
}

__device__ __forceinline__ int cell_linear(const Grid& g, int cx, int cy, int cz) { return (cz * g.ny + cy) * g.nx + cx; }

// Squared lower bound on the distance from q to any indexed point OUTSIDE the explored box of
// Chebyshev radius r around q's cell (cx,cy,cz).  Faces that coincide with the grid border have
// nothing behind them (border cells hold the clamped outliers) and are ignored -> +inf when the
// explored box covers the whole grid.
__device__ __forceinline__ float unexplored_bound_sq(const Grid& g, float qx, float qy, float qz, int cx, int cy, int cz, int r) {
  float best = 3.0e38f;
  const float h = g.h;
  if (cx - r > 0) best = fminf(best, qx - (g.ox + (float)(cx - r) * h));
  if (cx + r < g.nx - 1) best = fminf(best, (g.ox + (float)(cx + r + 1) * h) - qx);
  if (cy - r > 0) best = fminf(best, qy - (g.oy + (float)(cy - r) * h));
  if (cy + r < g.ny - 1) best = fminf(best, (g.oy + (float)(cy + r + 1) * h) - qy);
  if (cz - r > 0) best = fminf(best, qz - (g.oz + (float)(cz - r) * h));
  if (cz + r < g.nz - 1) best = fminf(best, (g.oz + (float)(cz + r + 1) * h) - qz);
  if (best > 1.0e38f) return 3.4028234664e38f;  // everything explored: FLT_MAX ends the search even when nothing was found
  best = fmaxf(best - g.slack, 0.0f);
  return best * best;
}

// lower bound of the squared distance from a query (cell row cy, cz) to anything in grid row (ry, rz): the (y,z) gap to its cells
__device__ __forceinline__ float row_gap_sq(const Grid& g, int ry, int rz, int cy, int cz, float qy, float qz) {
  float gy = 0.f, gz = 0.f;
  if (ry > cy) gy = (g.oy + (float)ry * g.h) - qy; else if (ry < cy) gy = qy - (g.oy + (float)(ry + 1) * g.h);
  if (rz > cz) gz = (g.oz + (float)rz * g.h) - qz; else if (rz < cz) gz = qz - (g.oz + (float)(rz + 1) * g.h);
  gy = fmaxf(gy - g.slack, 0.f);
  gz = fmaxf(gz - g.slack, 0.f);
  return gy * gy + gz * gz;
}

// --- per-cell (y,z) extents ---------------------------------------------------------------------
// A cell's points usually fill a small part of its (y,z) square: a scan line, a wall seen along x, the ground.  One 32-bit word per
// cell holds their extent inside the cell, y and z, in 1/255 of the cell edge, rounded outwards: {y lo, y hi, z lo, z hi}, a byte each;
// an empty cell has lo = 255 > hi = 0.  A query's distance to what a run of cells really holds - instead of to their squares - drops
// (query, row) pairs without a memory access and, above all, shortens the walks along structures parallel to x: the x-reach of a walk is
// sqrt(bound - gap), and the squares' gap is up to a cell edge short of the true one.
constexpr unsigned int kCellBoxEmpty = 0x00ff00ffu;
__device__ __forceinline__ unsigned int cell_box_union(unsigned int a, unsigned int b) {  // (an empty box is the neutral element)
  const unsigned int ylo = min(a & 0xffu, b & 0xffu), yhi = max((a >> 8) & 0xffu, (b >> 8) & 0xffu);
  const unsigned int zlo = min((a >> 16) & 0xffu, (b >> 16) & 0xffu), zhi = max(a >> 24, b >> 24);
  return ylo | (yhi << 8) | (zlo << 16) | (zhi << 24);
}
// lower bound of the squared (y,z) distance from (qy, qz) to any point of the cells of grid row (ry, rz) whose boxes were united in `box`
// (same slack as row_gap_sq: a point may sit a rounding error outside the cell it was assigned to); 3.4e38 for an empty box
__device__ __forceinline__ float box_gap_sq(const Grid& g, int ry, int rz, unsigned int box, float qy, float qz) {
  const float ylo = (float)(box & 0xffu), yhi = (float)((box >> 8) & 0xffu), zlo = (float)((box >> 16) & 0xffu), zhi = (float)(box >> 24);
  if (ylo > yhi) return 3.4028234664e38f;
  const float s = g.h * (1.0f / 255.0f);
  const float y0 = g.oy + (float)ry * g.h, z0 = g.oz + (float)rz * g.h;
  float gy = fmaxf(fmaxf((y0 + ylo * s) - qy, qy - (y0 + yhi * s)), 0.f);
  float gz = fmaxf(fmaxf((z0 + zlo * s) - qz, qz - (z0 + zhi * s)), 0.f);
  gy = fmaxf(gy - g.slack, 0.f);
  gz = fmaxf(gz - g.slack, 0.f);
  return gy * gy + gz * gz;
}
__global__ void k_fill_u32(unsigned int* a, unsigned int* b, int n, unsigned int v) {  // the frame of a padded table: n entries at a, n at b
  const int t = threadIdx.x;
  if (t < n) a[t] = v; else if (t < 2 * n) b[t - n] = v;
}
// one thread per cell over the cell-sorted points (cells with more than kCellBoxScan points keep the whole square: bounded time)
constexpr int kCellBoxScan = 512;
__global__ void __launch_bounds__(256) k_cell_boxes(const float4* __restrict__ pts, const int* __restrict__ cell_start, Grid g, unsigned int* __restrict__ box) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= g.ncells) return;
  const int s = cell_start[c], e = cell_start[c + 1];
  unsigned int out = kCellBoxEmpty;
  if (e > s) {
    out = 0xff00ff00u;  // the whole square
    if (e - s <= kCellBoxScan) {
      const int cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
      const float y0 = g.oy + (float)cy * g.h, z0 = g.oz + (float)cz * g.h;
      float ymin = 3.0e38f, ymax = -3.0e38f, zmin = 3.0e38f, zmax = -3.0e38f;
      for (int i = s; i < e; ++i) {
        const float4 p = pts[i];
        ymin = fminf(ymin, p.y); ymax = fmaxf(ymax, p.y);
        zmin = fminf(zmin, p.z); zmax = fmaxf(zmax, p.z);
      }
      // outwards, with one unit to spare for the float arithmetic on either side of the encoding
      const float k = 255.0f * g.inv_h;
      const int ylo = max(0, min(255, (int)floorf((ymin - y0) * k) - 1)), yhi = max(0, min(255, (int)ceilf((ymax - y0) * k) + 1));
      const int zlo = max(0, min(255, (int)floorf((zmin - z0) * k) - 1)), zhi = max(0, min(255, (int)ceilf((zmax - z0) * k) + 1));
      out = (unsigned int)ylo | ((unsigned int)yhi << 8) | ((unsigned int)zlo << 16) | ((unsigned int)zhi << 24);
    }
  }
  box[c] = out;
}

// ---------------------------------------------------------------------------------------------
// Build kernels
// ---------------------------------------------------------------------------------------------

// raw strided host layout (already copied to the device) -> float4 {x,y,z,index}; per-block bbox
// partials bbox_part[block][8] = {min x,y,z, max x,y,z, #non-finite points, 0} (reduced on the host: no contended atomics).
__global__ void __launch_bounds__(256) k_unpack_bbox(const unsigned char* __restrict__ raw, size_t stride_bytes, int n, float4* __restrict__ pts,
                                                      float* __restrict__ bbox_part) {
  __shared__ float lds[4][7];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  float bad = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = reinterpret_cast<const float*>(raw + (size_t)i * stride_bytes);
    float x = p[0], y = p[1], z = p[2];
    pts[i] = make_float4(x, y, z, __int_as_float(i));
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) bad += 1.f;
    mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
    mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
    mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lds[wave][d] = mn[d];
      lds[wave][3 + d] = mx[d];
    }
    lds[wave][6] = bad;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int d = threadIdx.x;
    float v = lds[0][d];
    for (int w = 1; w < 4; ++w) v = d < 3 ? fminf(v, lds[w][d]) : (d < 6 ? fmaxf(v, lds[w][d]) : v + lds[w][d]);
    bbox_part[blockIdx.x * 8 + d] = v;
  }
}

// histogram of points per cell; keys[i] = linear cell of point i, rank[i] = how many points of that cell arrived before it (the
// returned value of the atomic: the scatter then needs no second round of atomics; the arrival order is arbitrary and is replaced by
// k_cell_rank's (x, index) order).  stride > 1: only every stride-th point is counted (occupancy estimates while the voxel edge is
// being chosen; no keys / ranks are written then).
__global__ void __launch_bounds__(256) k_cell_count(const float4* __restrict__ pts, int n, Grid g, int* __restrict__ keys, int* __restrict__ counts, int* __restrict__ rank,
                                                     int stride) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; (long long)t * stride < n; t += gridDim.x * blockDim.x) {
    const int i = t * stride;
    float4 p = pts[i];
    int cx, cy, cz;
    cell_coords(g, p.x, p.y, p.z, cx, cy, cz);
    int c = cell_linear(g, cx, cy, cz);
    if (stride == 1) {
      keys[i] = c;
      rank[i] = atomicAdd(&counts[c], 1);
    } else {
      atomicAdd(&counts[c], 1);
    }
  }
}

// Exclusive scan over `n` ints, 3 kernels, 4096 elements per block.
// Also accumulates the occupancy statistic sum(count^2) through per-tile partials (no atomics).
constexpr int kScanBlock = 256;
constexpr int kScanPerThread = 16;
constexpr int kScanTile = kScanBlock * kScanPerThread;

__device__ __forceinline__ int block_exclusive_scan(int v, int* lds /*>= 4 + 1 ints*/, int& block_total) {
  // inclusive wave scan
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int wave_off = 0;
  int total = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; ++w) {
    int s = lds[w];
    if (w < wave) wave_off += s;
    total += s;
  }
  __syncthreads();
  block_total = total;
  return wave_off + inc - v;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_tiles(const int* __restrict__ counts, int n, int* __restrict__ tile_sums,
                                                           unsigned long long* __restrict__ tile_sq /*[ntiles] sum of count^2, or null*/) {
  __shared__ int lds[8];
  __shared__ unsigned long long lsq[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
  int s = 0;
  unsigned long long sq = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    int i = base + j;
    int c = (i < n) ? counts[i] : 0;
    s += c;
    sq += (unsigned long long)c * (unsigned long long)c;
  }
  int total;
  block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
  if (tile_sq) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) lsq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) tile_sq[blockIdx.x] = lsq[0] + lsq[1] + lsq[2] + lsq[3];
  }
}

// single block: exclusive scan of tile_sums in place (ntiles <= 1024 * per)
__global__ void __launch_bounds__(kScanBlock) k_scan_tile_sums(int* __restrict__ tile_sums, int ntiles, const unsigned long long* __restrict__ tile_sq,
                                                               unsigned long long* __restrict__ occ_out /*[1] sum of count^2*/) {
  __shared__ int lds[8];
  __shared__ int carry_s;
  __shared__ unsigned long long lsq[4];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  unsigned long long sq = 0;
  for (int base = 0; base < ntiles; base += kScanBlock) {
    int i = base + threadIdx.x;
    int v = (i < ntiles) ? tile_sums[i] : 0;
    if (tile_sq && i < ntiles) sq += tile_sq[i];
    int total;
    int ex = block_exclusive_scan(v, lds, total);
    int carry = carry_s;
    if (i < ntiles) tile_sums[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (tile_sq) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) lsq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) occ_out[0] = lsq[0] + lsq[1] + lsq[2] + lsq[3];
  }
}

// The cell-start table is framed by kCellPad entries on each side (0 in front, the total behind): the four bounds around any cell
// are then one 16-byte load (see k_gicp_pass).
constexpr int kCellPad = 8;  // (k_gicp_pass_st fetches the starts of eight consecutive cells with two 16-byte loads)
// out[i] = exclusive prefix of counts (out has n + 1 entries; out[n] = total, repeated kCellPad times behind it)
__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const int* __restrict__ counts, int n, const int* __restrict__ tile_offsets, int* __restrict__ out) {
  __shared__ int lds[8];
  const int base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
  int c[kScanPerThread];
  int s = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    int i = base + j;
    c[j] = (i < n) ? counts[i] : 0;
    s += c[j];
  }
  int total;
  int ex = block_exclusive_scan(s, lds, total) + tile_offsets[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    int i = base + j;
    if (i < n) out[i] = ex;
    ex += c[j];
    if (i == n - 1) {
#pragma unroll
      for (int t = 0; t <= kCellPad; ++t) out[n + t] = ex;
    }
  }
}

// The tile pipeline needs two exclusive scans over the same histogram: of the tile counts (where a tile's queries start) and of the
// tiles' batch counts, ceil(count / 32) (where its batches start).  Both ride in ONE scan of packed 64-bit values (count in the low
// word, batches in the high word: neither sum reaches 2^31), three launches instead of seven.
__device__ __forceinline__ unsigned long long pack_tile(int count) {
  return (unsigned long long)(unsigned int)count | ((unsigned long long)(unsigned int)((count + 31) >> 5) << 32);
}
__device__ __forceinline__ unsigned long long block_exclusive_scan64(unsigned long long v, unsigned long long* lds /*>= 4*/, unsigned long long& block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  unsigned long long wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; ++w) {
    const unsigned long long s = lds[w];
    if (w < wave) wave_off += s;
    total += s;
  }
  __syncthreads();
  block_total = total;
  return wave_off + inc - v;
}
__global__ void __launch_bounds__(kScanBlock) k_scan2_tiles(const int* __restrict__ counts, int n, unsigned long long* __restrict__ tile_sums) {
  __shared__ unsigned long long lds[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
  unsigned long long s = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) s += (base + j < n) ? pack_tile(counts[base + j]) : 0ull;
  unsigned long long total;
  block_exclusive_scan64(s, lds, total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kScanBlock) k_scan2_tile_sums(unsigned long long* __restrict__ tile_sums, int ntiles) {  // single block, in place
  __shared__ unsigned long long lds[4];
  __shared__ unsigned long long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += kScanBlock) {
    const int i = base + threadIdx.x;
    const unsigned long long v = (i < ntiles) ? tile_sums[i] : 0ull;
    unsigned long long total;
    const unsigned long long ex = block_exclusive_scan64(v, lds, total);
    const unsigned long long carry = carry_s;
    if (i < ntiles) tile_sums[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
}
// start[i] / bstart[i] = exclusive prefixes of count / ceil(count / 32); start has n + 1 + kCellPad entries (as k_scan_apply writes
// them), *n_batches = the total number of batches
__global__ void __launch_bounds__(kScanBlock) k_scan2_apply(const int* __restrict__ counts, int n, const unsigned long long* __restrict__ tile_offsets, int* __restrict__ start,
                                                            int* __restrict__ bstart, int* __restrict__ n_batches) {
  __shared__ unsigned long long lds[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
  unsigned long long c[kScanPerThread];
  unsigned long long s = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    c[j] = (base + j < n) ? pack_tile(counts[base + j]) : 0ull;
    s += c[j];
  }
  unsigned long long total;
  unsigned long long ex = block_exclusive_scan64(s, lds, total) + tile_offsets[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) {
    const int i = base + j;
    if (i < n) {
      start[i] = (int)(unsigned int)ex;
      bstart[i] = (int)(ex >> 32);
    }
    ex += c[j];
    if (i == n - 1) {
#pragma unroll
      for (int t = 0; t <= kCellPad; ++t) start[n + t] = (int)(unsigned int)ex;
      bstart[n] = (int)(ex >> 32);
      *n_batches = (int)(ex >> 32);
    }
  }
}

// scatter into cell segments (arrival order inside a cell: see k_cell_count)
__global__ void __launch_bounds__(256) k_cell_scatter(const float4* __restrict__ pts, const int* __restrict__ keys, const int* __restrict__ rank, int n,
                                                       const int* __restrict__ cell_start, float4* __restrict__ tmp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) tmp[cell_start[keys[i]] + rank[i]] = pts[i];
}

// make the order inside each cell deterministic AND useful: rank by (x, original index).  Cells of a row are
// ascending in x, so every (y,z) row of the cell-sorted cloud ends up fully sorted by x: the candidates of a row
// within |x - qx| <= r are one contiguous sub-range that bisection finds
__global__ void __launch_bounds__(256) k_cell_rank(const float4* __restrict__ tmp, int n, Grid g, const int* __restrict__ cell_start, float4* __restrict__ sorted,
                                                    int* __restrict__ perm) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    float4 me = tmp[p];
    int cx, cy, cz;
    cell_coords(g, me.x, me.y, me.z, cx, cy, cz);
    int c = cell_linear(g, cx, cy, cz);
    int s = cell_start[c], e = cell_start[c + 1];
    int my = __float_as_int(me.w);
    int rank = 0;
    for (int q = s; q < e; ++q) {
      const float4 o = tmp[q];
      rank += (o.x < me.x) || (o.x == me.x && __float_as_int(o.w) < my);
    }
    sorted[s + rank] = me;
    perm[s + rank] = my;
  }
}

// ---------------------------------------------------------------------------------------------
// Query order: the same points again, ordered by the Morton code of their 2^shift-cell tile (and by
// linear cell, then sorted position, inside a tile).  Batches of consecutive queries are then compact
// 3-D blobs, which is what lets a wave stage ONE small target region in LDS for the whole batch.
// qpts[i] = {x, y, z, bitcast(sorted position)}.  Built without a sort: a point's destination is
// tile_start[tile] + (points of the tile's cells that precede its cell) + (rank inside its cell).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int spread3(unsigned int v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__device__ __forceinline__ unsigned int tile_key(int cx, int cy, int cz, int shift) {
  return spread3((unsigned)(cx >> shift)) | (spread3((unsigned)(cy >> shift)) << 1) | (spread3((unsigned)(cz >> shift)) << 2);
}

// The points arrive sorted by cell, so the 64 points of a wave fall into a handful of tiles: one atomic per RUN of equal
// keys (run heads found with a shuffle + ballot) instead of one per point (same-address atomics serialise at ~11 ns).
__global__ void __launch_bounds__(256) k_tile_count(const float4* __restrict__ sorted, int n, Grid g, int shift, int* __restrict__ tile_counts) {
  const int lane = threadIdx.x & 63;
  for (int base = blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n; base += gridDim.x * blockDim.x) {
    const int p = base + lane;
    int key = -1;
    if (p < n) {
      const float4 q = sorted[p];
      int cx, cy, cz;
      cell_coords(g, q.x, q.y, q.z, cx, cy, cz);
      key = (int)tile_key(cx, cy, cz, shift);
    }
    const int prev = __shfl_up(key, 1);
    const bool head = lane == 0 || key != prev;
    const unsigned long long m = __ballot(head);
    if (head && key >= 0) {
      const unsigned long long higher = lane == 63 ? 0ull : (m >> (lane + 1)) << (lane + 1);
      const int next = higher ? __builtin_ctzll(higher) : 64;
      atomicAdd(&tile_counts[key], next - lane);
    }
  }
}

__global__ void __launch_bounds__(256) k_tile_place(const float4* __restrict__ sorted, int n, Grid g, int shift, const int* __restrict__ cell_start,
                                                     const int* __restrict__ tile_start, float4* __restrict__ qpts) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const float4 q = sorted[p];
    int cx, cy, cz;
    cell_coords(g, q.x, q.y, q.z, cx, cy, cz);
    const int ts = 1 << shift;
    const int x0 = (cx >> shift) << shift, y0 = (cy >> shift) << shift, z0 = (cz >> shift) << shift;
    const int x1 = min(x0 + ts, g.nx);  // exclusive
    int before = 0;
    for (int z = z0; z <= cz; ++z) {
      const int ylast = (z == cz) ? cy : min(y0 + ts, g.ny) - 1;
      for (int y = y0; y <= ylast; ++y) {
        const int row = (z * g.ny + y) * g.nx;
        const int xe = (z == cz && y == cy) ? cx : x1;  // own row: only the cells before the own cell
        before += cell_start[row + xe] - cell_start[row + x0];
      }
    }
    const int own = cell_start[(cz * g.ny + cy) * g.nx + cx];
    const int dst = tile_start[tile_key(cx, cy, cz, shift)] + before + (p - own);
    qpts[dst] = make_float4(q.x, q.y, q.z, __int_as_float(p));
  }
}

// The cell-sorted point array is framed by kSortedPad far-away sentinels on each side (see DeviceCloud::sorted).
constexpr int kSortedPad = 32;  // >= the widest walk window (NGICP_WALK_WINDOW) and >= 16 for the k-NN seed of k <= 32
__global__ void k_fill_sentinels(float4* __restrict__ padded, int n) {
  const int t = threadIdx.x;  // 2 * kSortedPad threads
  const float far = 3.0e38f;  // squared distance overflows to +inf: never closer than anything
  padded[t < kSortedPad ? t : n + t] = make_float4(far, far, far, __int_as_float(-1));
}

// Query batches: runs of at most kBatchQueries consecutive entries of qpts that never cross a tile
// boundary, so a batch's bounding box is at most one tile.  batch[b] = {first qpts index, count}.
constexpr int kBatchQueries = 32;

__global__ void __launch_bounds__(256) k_batch_fill(const int* __restrict__ tile_counts, const int* __restrict__ tile_start, const int* __restrict__ batch_start,
                                                     int nbins, int2* __restrict__ batches) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbins) return;
  const int cnt = tile_counts[t];
  const int b0 = batch_start[t], q0 = tile_start[t];
  for (int j = 0, done = 0; done < cnt; ++j, done += kBatchQueries) batches[b0 + j] = make_int2(q0 + done, min(kBatchQueries, cnt - done));
}

// Axis-aligned bounding box (metric, cloud frame) of every query batch: {cx, cy, cz, hx, hy, hz} centre and
// half extents.  The pass kernel transforms it with the trial pose instead of reducing over lanes.
__global__ void __launch_bounds__(256) k_batch_boxes(const float4* __restrict__ qpts, const int2* __restrict__ batches, const int* __restrict__ n_batches,
                                                      float* __restrict__ boxes6) {
  // a wave per batch, a lane per query (a thread per batch walked its 32 queries one dependent load after the other)
  const int lane = threadIdx.x & 63, nb = *n_batches;
  for (int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < nb; b += (gridDim.x * blockDim.x) >> 6) {
  const int2 bd = batches[b];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int j = lane; j < bd.y; j += 64) {
    const float4 p = qpts[bd.x + j];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
    }
  if (lane == 0) {
    float* o = boxes6 + (size_t)b * 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      o[d] = 0.5f * (mn[d] + mx[d]);
      o[3 + d] = 0.5f * (mx[d] - mn[d]);
    }
  }
  }
}

}  // namespace ngk
