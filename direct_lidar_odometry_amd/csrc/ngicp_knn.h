// Exact k-nearest-neighbour search on the voxel grid + per-point covariance estimation.
// Replaces KdTreeFLANN::nearestKSearch (/root/reference/include/nano_gicp/nanoflann.hpp:141-152,
// impl/nanoflann_impl.hpp:1230-1250,1355-1418) and NanoGICP::calculate_covariances
// (impl/nano_gicp_impl.hpp:300-357).
//
// Exactness: rings of cells around the query cell are searched until the k-th best squared
// distance is <= the squared distance to the nearest unexplored cell face (eps = 0, like the
// reference).  Distances are float32 ((dx*dx + dy*dy) + dz*dz, no FMA: this TU is compiled with
// -ffp-contract=off) exactly as impl/nanoflann_impl.hpp:441-449.  Ties are broken by visiting
// order with strict '<' like KNNResultSet::addPoint (impl/nanoflann_impl.hpp:184-211); the visiting
// order differs from a kd-tree's, so among EXACTLY equal distances a different index may win
// (SURVEY.md §7 "Ties").
#pragma once
#include <type_traits>

#include "ngicp_grid.h"
#include "ngicp_math.h"

namespace ngk {

// Sorted (ascending) top-K list held in NAMED scalars (a recursive struct, one level per slot): an array member, even with
// compile-time indices after unrolling, was left in private memory by the compiler at K >= 20 (164 bytes of scratch per lane,
// every insert a round of scratch loads and stores) and promoted to LDS at K = 10.  TopK<K> = slots 0..K-2 (`head`) + slot K-1.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int K>
struct TopK {
  TopK<K - 1> head;
  float d_;
  int id_;
  __device__ __forceinline__ void init() {
    head.init();
    d_ = 3.4028234664e38f;  // FLT_MAX, KNNResultSet::init (impl/nanoflann_impl.hpp:168-174)
    id_ = -1;
  }
  __device__ __forceinline__ float last_d() const { return d_; }
  __device__ __forceinline__ int last_id() const { return id_; }
  // insert (dist, index) behind the entries <= key (strict '>': entries equal to the key stay in front); an entry that does not
  // beat the last one changes nothing.  key == dist is the ordinary insert; key < 0 puts the entry in front of everything.
  __device__ __forceinline__ void insert(float key, float dist, int index) {
    const float pd = head.last_d();  // slot K-2 before it moves
    const int pi = head.last_id();
    const bool shift = pd > key;
    const bool here = !shift && (d_ > key);
    d_ = shift ? pd : (here ? dist : d_);
    id_ = shift ? pi : (here ? index : id_);
    head.insert(key, dist, index);
  }
  __device__ __forceinline__ float d_at(int s) const { return s == K - 1 ? d_ : head.d_at(s); }  // run-time slot: a select chain
  __device__ __forceinline__ float kth(int k) const { return d_at(k - 1); }
  template <int S>
  __device__ __forceinline__ float d() const {
    if constexpr (S == K - 1) return d_; else return head.template d<S>();
  }
  template <int S>
  __device__ __forceinline__ int id() const {
    if constexpr (S == K - 1) return id_; else return head.template id<S>();
  }
};
template <>
struct TopK<1> {
  float d_;
  int id_;
  __device__ __forceinline__ void init() {
    d_ = 3.4028234664e38f;
    id_ = -1;
  }
  __device__ __forceinline__ float last_d() const { return d_; }
  __device__ __forceinline__ int last_id() const { return id_; }
  __device__ __forceinline__ void insert(float key, float dist, int index) {
    if (d_ > key) {
      d_ = dist;
      id_ = index;
    }
  }
  __device__ __forceinline__ float d_at(int) const { return d_; }
  __device__ __forceinline__ float kth(int) const { return d_; }
  template <int S>
  __device__ __forceinline__ float d() const { return d_; }
  template <int S>
  __device__ __forceinline__ int id() const { return id_; }
};

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, const float4& p) {
  const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// Exact k-NN of (qx,qy,qz) in an indexed cloud, row by row.  Result: sorted positions, distances ascending.
//   rings   rings of cells around the query's cell are visited until the k-th best squared distance (`worst`) is <= the squared
//           distance to the nearest unexplored cell face (unexplored_bound_sq: eps = 0, like the reference);
//   rows    rings 0..1 are the 3 x 3 window of (y,z) rows, each the x-sorted run of the cells cx-1..cx+1; ring r >= 2 adds the
//           rows on the frame of the (2r+1)^2 window (runs cx-r..cx+r) and, for the rows inside the frame, the two end cells
//           cx-r and cx+r.  A row or cell whose gap to the query alone reaches `worst` is skipped without a memory access.
//           (A list-based bound only exists once the list holds k entries: until then `worst` is FLT_MAX and nothing is pruned.)
//   walk    a run is walked outward from a starting position (the query point itself when it belongs to the cloud: p >= 0; else
//           where qx sits in the run, interpolated), one window per round trip, right then left, each side only while
//           |dx|^2 + gap can still beat `worst`.
// Every position is visited at most once (no duplicates in the list).  Ties: the first visited stays in front (strict '<'), like
// KNNResultSet::addPoint (impl/nanoflann_impl.hpp:184-211); the visiting order is not a kd-tree's, so among EXACTLY equal
// distances another index may be kept (SURVEY.md §7 "Ties").
//
// TWO LANES PER QUERY (a pair: lanes 2i, 2i + 1; `sub` = lane & 1).  A thread per query left a 100k-point scan at 38 % of the
// chip's wave slots with every wave on its own dependency chain; a pair halves the chain instead of doubling the candidates:
//   * ONE sorted list per pair, slots 0..K/2-1 in lane 0, K/2..K-1 in lane 1.  An insert is a K/2-step predicated shift that
//     both lanes run at once: lane 0 shifts the candidate into its half and hands what falls off its end (or the candidate itself,
//     if it belongs behind) to lane 1 - the list, the k-th best and the tie order are exactly those of a single K-slot list;
//   * a window's eight points are fetched and distance-tested four per lane; passing candidates are marked per lane and taken in
//     increasing position (lane 0's marks, then lane 1's), each broadcast to the pair by a shuffle.
// Both lanes of a pair hold the same query and the same `worst`, so every branch below is uniform inside a pair.
template <int K>
struct PairTopK {
  static_assert(K % 2 == 0, "the list is split evenly over the two lanes of a pair");
  static constexpr int H = K / 2;
  TopK<H> part;  // slots sub * H .. sub * H + H - 1
  __device__ __forceinline__ void init() { part.init(); }
  // both lanes call with the same candidate
  __device__ __forceinline__ void insert(float dist, int index, int sub, int lane) {
    // what moves on to the upper half: the lower half's last entry if the candidate goes in front of it, else the candidate
    const float ld = part.last_d();
    const int li = part.last_id();
    const bool low = ld > dist;
    const float cd = __shfl(low ? ld : dist, lane & ~1);
    const int ci = __shfl(low ? li : index, lane & ~1);
    const float ck = __shfl(low ? -1.f : dist, lane & ~1);  // the lower half's old last entry stays in front of the upper half's entries
    part.insert(sub ? ck : dist, sub ? cd : dist, sub ? ci : index);  // (an entry that does not beat the half's last one changes nothing)
  }
  __device__ __forceinline__ float kth(int k, int sub, int lane) const {  // the same in both lanes
    const int slot = k - 1;
    return __shfl(part.d_at(slot >= H ? slot - H : slot), (lane & ~1) | (slot >= H ? 1 : 0));
  }
};

// WL = points of a window per lane (a window = 2 WL points = one round trip of the pair).  Measured (k = 20): 4 is best when the
// launch fills the chip (250k OS1 points: 0.249 ms against 0.283 / 0.299 ms with 6 / 8), 6 when it does not and a wave's chain of
// round trips is what counts (100k-point VLP-16 scan, 61 % of the wave slots: 0.119 against 0.128 ms); the host picks by cloud size.
// The lane's points of a window (positions w + WL sub ...; `valid`: which of the window's points belong to the run).
template <int K, int WL>
__device__ __forceinline__ void knn_take_window(const float4 (&c)[WL], int w, unsigned int valid, float qx, float qy, float qz, int k, PairTopK<K>& top, float& worst,
                                                int sub, int lane) {
  float d[WL];
  unsigned int mask = 0;
#pragma unroll
  for (int j = 0; j < WL; ++j) {
    d[j] = sqdist(qx, qy, qz, c[j]);
    if (((valid >> (WL * sub + j)) & 1u) && d[j] < worst) mask |= 1u << j;
  }
#pragma unroll
  for (int owner = 0; owner < 2; ++owner) {
    unsigned int m = __shfl(mask, (lane & ~1) | owner);  // (uniform in the pair)
    while (m) {
      const int j = __ffs((int)m) - 1;
      m &= m - 1;
      float dj = d[0];
#pragma unroll
      for (int t = 1; t < WL; ++t) dj = (j == t) ? d[t] : dj;
      dj = __shfl(dj, (lane & ~1) | owner);
      if (dj < worst) {  // (the bound may have tightened since the candidate was marked)
        top.insert(dj, w + WL * owner + j, sub, lane);
        worst = fminf(worst, top.kth(k, sub, lane));
      }
    }
  }
}

template <int K, int WL>
__device__ __forceinline__ void knn_walk_row(const float4* __restrict__ sorted, int s, int e, int m, float qx, float qy, float qz, float gap, int k, PairTopK<K>& top,
                                             float& worst, int sub, int lane) {
  constexpr int kKnnW = 2 * WL;
  constexpr unsigned int kAll = (1u << kKnnW) - 1u;
  m = min(max(m, s), e - 1);
  for (int w = m; w < e; w += kKnnW) {  // [m, e)
    float4 c[WL];
#pragma unroll
    for (int j = 0; j < WL; ++j) c[j] = sorted[min(w + WL * sub + j, e - 1)];
    knn_take_window<K, WL>(c, w, w + kKnnW <= e ? kAll : (kAll >> (w + kKnnW - e)), qx, qy, qz, k, top, worst, sub, lane);
    const float dr = __shfl(c[WL - 1].x, lane | 1) - qx;  // the window's (or the run's) last point: everything beyond has a larger x
    if (dr > 0.f && dr * dr + gap > worst) break;
  }
  for (int w = m - kKnnW; w + kKnnW > s; w -= kKnnW) {  // [s, m), nearest window first
    float4 c[WL];
#pragma unroll
    for (int j = 0; j < WL; ++j) c[j] = sorted[max(w + WL * sub + j, s)];
    knn_take_window<K, WL>(c, w, w >= s ? kAll : ((kAll << (s - w)) & kAll), qx, qy, qz, k, top, worst, sub, lane);
    const float dl = qx - __shfl(c[0].x, lane & ~1);
    if (dl > 0.f && dl * dl + gap > worst) break;
  }
}

constexpr int kKnnBlock = 128;             // threads per block of the k-NN kernels
constexpr int kKnnPairs = kKnnBlock / 2;   // queries per block (each pair parks 36 row bounds in LDS)

template <int K, int WL>
__device__ __forceinline__ void knn_search(const Grid& g, const float4* __restrict__ sorted, const int* __restrict__ cell_start, float qx, float qy, float qz,
                                           int p, int k, PairTopK<K>& top, int* __restrict__ lds_bounds /* [36][kKnnPairs], this pair's column */, int sub, int lane) {
  top.init();
  float worst = 3.4028234664e38f;
  int cx, cy, cz;
  cell_coords(g, qx, qy, qz, cx, cy, cz);
  // ---- rings 0..1: the bounds of all nine rows of the 3 x 3 window are fetched in ONE round trip (one 16-byte load per row gives
  //      the starts of the cells cx-1, cx, cx+1, cx+2: the cell-start table is padded for it; lane 0 fetches rows 0..4, lane 1 rows
  //      5..8) and parked in LDS; the query's own row is walked first, from the query itself when it is a point of this cloud, else
  //      from where qx sits inside its own cell ----
  {
    struct alignas(4) Bounds4 { int v[4]; };
    {
      Bounds4 bnd[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int t = min(5 * sub + u, 8);
        const int y = min(max(cy + t % 3 - 1, 0), g.ny - 1), z = min(max(cz + t / 3 - 1, 0), g.nz - 1);  // (rows outside the grid are skipped below)
        bnd[u] = *reinterpret_cast<const Bounds4*>(cell_start + ((z * g.ny + y) * g.nx + cx - 1));
      }
      // (all five in flight before the first is parked: the compiler had put the fifth load behind the waits for the other four - one more
      // dependent round trip per query)
      asm volatile("" ::"v"(bnd[0].v[0]), "v"(bnd[1].v[0]), "v"(bnd[2].v[0]), "v"(bnd[3].v[0]), "v"(bnd[4].v[0]));
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int t = 5 * sub + u;
        if (t < 9)
#pragma unroll
          for (int v = 0; v < 4; ++v) lds_bounds[(t * 4 + v) * kKnnPairs] = bnd[u].v[v];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the pair's other lane reads what this one parked)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float fx = fminf(fmaxf((qx - (g.ox + (float)cx * g.h)) * g.inv_h, 0.f), 1.f);
    for (int o = 0; o < 9; ++o) {
      const int t = o == 0 ? 4 : (o <= 4 ? o - 1 : o);  // the own row first
      const int y = cy + t % 3 - 1, z = cz + t / 3 - 1;
      if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
      const float gap = t == 4 ? 0.f : row_gap_sq(g, y, z, cy, cz, qy, qz);
      if (gap >= worst) continue;
      const int v0 = lds_bounds[(t * 4 + 0) * kKnnPairs], v1 = lds_bounds[(t * 4 + 1) * kKnnPairs], v2 = lds_bounds[(t * 4 + 2) * kKnnPairs],
                v3 = lds_bounds[(t * 4 + 3) * kKnnPairs];
      const int s = cx > 0 ? v0 : v1, e = cx < g.nx - 1 ? v3 : v2;
      if (e <= s) continue;
      const int m = (t == 4 && p >= 0) ? p : v1 + (int)(fx * (float)(v2 - v1));
      knn_walk_row<K, WL>(sorted, s, e, m, qx, qy, qz, gap, k, top, worst, sub, lane);
    }
  }
  // ---- rings 2, 3, ... while the k-th best is not provably exact: the rows on the frame of the (2r+1)^2 window (runs cx-r..cx+r)
  //      and, for the rows inside the frame, the two end cells cx-r and cx+r ----
  const int rmax = max(max(g.nx, g.ny), g.nz);
  for (int r = 2; r <= rmax + 1; ++r) {
    if (worst <= unexplored_bound_sq(g, qx, qy, qz, cx, cy, cz, r - 1)) break;
    const int xa = max(cx - r, 0), xb = min(cx + r, g.nx - 1);
    const float frac = fminf(fmaxf((qx - (g.ox + (float)xa * g.h)) / ((float)(xb + 1 - xa) * g.h), 0.f), 1.f);
    // lower bounds of |dx| to the two end cells of an inner row (cells cx-r and cx+r)
    const float xl = fmaxf(qx - (g.ox + (float)(cx - r + 1) * g.h) - g.slack, 0.f), xr = fmaxf((g.ox + (float)(cx + r) * g.h) - qx - g.slack, 0.f);
    const int side = 2 * r + 1;
    for (int t = 0; t < side * side; ++t) {
      const int dz = t / side - r, dy = t % side - r;
      const int y = cy + dy, z = cz + dz;
      if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
      const float gap = row_gap_sq(g, y, z, cy, cz, qy, qz);
      if (gap >= worst) continue;
      const int row = (z * g.ny + y) * g.nx;
      const bool full = dz == -r || dz == r || dy == -r || dy == r;  // else an inner row: only its two end cells are new
      for (int u = 0; u < (full ? 1 : 2); ++u) {
        int c0, c1, m_hint;  // cells [c0, c1] of the row; where to start (0: interpolate, 1: the run's last point, 2: its first)
        if (full) {
          c0 = xa; c1 = xb; m_hint = 0;
        } else if (u == 0) {
          if (cx - r < 0 || xl * xl + gap >= worst) continue;
          c0 = c1 = cx - r; m_hint = 1;
        } else {
          if (cx + r > g.nx - 1 || xr * xr + gap >= worst) continue;
          c0 = c1 = cx + r; m_hint = 2;
        }
        const int s = cell_start[row + c0], e = cell_start[row + c1 + 1];
        if (e <= s) continue;
        const int m = m_hint == 1 ? e - 1 : (m_hint == 2 ? s : s + (int)(frac * (float)(e - s)));
        knn_walk_row<K, WL>(sorted, s, e, m, qx, qy, qz, gap, k, top, worst, sub, lane);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1: covariance of every point of an indexed cloud (thread per point, sorted order)
// covs6[i] = {xx,xy,xz,yy,yz,zz} of the regularised 3x3 (FP64), i = sorted position.
// ---------------------------------------------------------------------------------------------
enum { REG_NONE = 0, REG_MIN_EIG = 1, REG_NORMALIZED_MIN_EIG = 2, REG_PLANE = 3, REG_FROBENIUS = 4 };

template <int K, int WL>
__global__ void __launch_bounds__(kKnnBlock) k_covariances(const float4* __restrict__ sorted, const int* __restrict__ cell_start, Grid g, int first, int n, int k, int reg,
                                                            double* __restrict__ covs6) {
  __shared__ int lds_bounds[36 * kKnnPairs];
  constexpr int H = K / 2;
  const int lane = threadIdx.x & 63, sub = threadIdx.x & 1, pair = threadIdx.x >> 1;
  const int i = first + blockIdx.x * kKnnPairs + pair;  // a pair of lanes per point; points [first, n) of the sorted cloud (a rank's block when K1 is sharded)
  if (i >= n) return;
  const float4 q = sorted[i];
  PairTopK<K> top;
  knn_search<K, WL>(g, sorted, cell_start, q.x, q.y, q.z, i, k, top, lds_bounds + pair, sub, lane);

  // impl/nano_gicp_impl.hpp:315-321: mean-centre the k neighbours (FP64), C = X X^T / k.  The neighbours are added in list order:
  // lane 0 adds its slots 0..K/2-1, lane 1 takes the sums over and continues with K/2..k-1 (the association of a single thread).
  float4 nb[H];
  static_for<0, H>([&](auto S) {
    const int id = top.part.template id<S.value>();
    nb[S.value] = (sub * H + S.value < k && id >= 0) ? sorted[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  });
  double mx = 0, my = 0, mz = 0;
  if (sub == 0) {
#pragma unroll
    for (int j = 0; j < H; ++j)
      if (j < k) {
        mx += (double)nb[j].x;
        my += (double)nb[j].y;
        mz += (double)nb[j].z;
      }
  }
  mx = __shfl(mx, lane & ~1);
  my = __shfl(my, lane & ~1);
  mz = __shfl(mz, lane & ~1);
  if (sub == 1) {
#pragma unroll
    for (int j = 0; j < H; ++j)
      if (H + j < k) {
        mx += (double)nb[j].x;
        my += (double)nb[j].y;
        mz += (double)nb[j].z;
      }
  }
  mx = __shfl(mx, lane | 1) / (double)k;
  my = __shfl(my, lane | 1) / (double)k;
  mz = __shfl(mz, lane | 1) / (double)k;
  double C[6] = {0, 0, 0, 0, 0, 0};
  auto add_centred = [&](int first) {
#pragma unroll
    for (int j = 0; j < H; ++j)
      if (first + j < k) {
        const double x = (double)nb[j].x - mx, y = (double)nb[j].y - my, z = (double)nb[j].z - mz;
        C[0] += x * x; C[1] += x * y; C[2] += x * z;
        C[3] += y * y; C[4] += y * z; C[5] += z * z;
      }
  };
  if (sub == 0) add_centred(0);
#pragma unroll
  for (int e = 0; e < 6; ++e) C[e] = __shfl(C[e], lane & ~1);
  if (sub == 0) return;  // lane 1 finishes the point
  add_centred(H);
#pragma unroll
  for (int e = 0; e < 6; ++e) C[e] = C[e] / (double)k;

  double out[6];
  if (reg == REG_NONE) {  // :323-324
#pragma unroll
    for (int e = 0; e < 6; ++e) out[e] = C[e];
  } else if (reg == REG_FROBENIUS) {  // :325-330
    double Cl[6] = {C[0] + 1e-3, C[1], C[2], C[3] + 1e-3, C[4], C[5] + 1e-3};
    double Ci[6];
    inv3_sym(Cl, Ci);
    double nrm = Ci[0] * Ci[0] + Ci[3] * Ci[3] + Ci[5] * Ci[5] + 2.0 * (Ci[1] * Ci[1] + Ci[2] * Ci[2] + Ci[4] * Ci[4]);
    nrm = sqrt(nrm);
#pragma unroll
    for (int e = 0; e < 6; ++e) Ci[e] = Ci[e] / nrm;
    inv3_sym(Ci, out);
  } else {  // :331-353  SVD path (PLANE is the only mode DLO exercises)
    double w[3], V[9];
    eig3_sym(C, w, V);
    const double a0 = fabs(w[0]), a1 = fabs(w[1]), a2 = fabs(w[2]);
    double v0, v1, v2;
    if (reg == REG_PLANE) {
      // singular values sorted descending get (1, 1, 1e-3): the smallest one gets 1e-3
      const int imin = (a0 <= a1 && a0 <= a2) ? 0 : ((a1 <= a2) ? 1 : 2);
      v0 = imin == 0 ? 1e-3 : 1.0;
      v1 = imin == 1 ? 1e-3 : 1.0;
      v2 = imin == 2 ? 1e-3 : 1.0;
    } else if (reg == REG_MIN_EIG) {
      v0 = fmax(a0, 1e-3); v1 = fmax(a1, 1e-3); v2 = fmax(a2, 1e-3);
    } else {
      const double m = fmax(a0, fmax(a1, a2));
      v0 = fmax(a0 / m, 1e-3); v1 = fmax(a1 / m, 1e-3); v2 = fmax(a2 / m, 1e-3);
    }
    // U diag(v) V^T with U == V (columns of V are eigenvectors)
    out[0] = V[0] * v0 * V[0] + V[1] * v1 * V[1] + V[2] * v2 * V[2];
    out[1] = V[0] * v0 * V[3] + V[1] * v1 * V[4] + V[2] * v2 * V[5];
    out[2] = V[0] * v0 * V[6] + V[1] * v1 * V[7] + V[2] * v2 * V[8];
    out[3] = V[3] * v0 * V[3] + V[4] * v1 * V[4] + V[5] * v2 * V[5];
    out[4] = V[3] * v0 * V[6] + V[4] * v1 * V[7] + V[5] * v2 * V[8];
    out[5] = V[6] * v0 * V[6] + V[7] * v1 * V[7] + V[8] * v2 * V[8];
  }
  double* o = covs6 + (size_t)i * 6;
#pragma unroll
  for (int e = 0; e < 6; ++e) o[e] = out[e];
}

// Test hook: exact kNN of arbitrary queries (float4 xyz_) in an indexed cloud; outputs ORIGINAL indices.
template <int K>
__global__ void __launch_bounds__(kKnnBlock) k_knn_queries(const float4* __restrict__ sorted, const int* __restrict__ cell_start, Grid g, const float4* __restrict__ queries,
                                                      int nq, int k, int* __restrict__ out_idx, float* __restrict__ out_d2) {
  __shared__ int lds_bounds[36 * kKnnPairs];
  constexpr int H = K / 2;
  const int lane = threadIdx.x & 63, sub = threadIdx.x & 1, pair = threadIdx.x >> 1;
  const int i = blockIdx.x * kKnnPairs + pair;
  if (i >= nq) return;
  const float4 q = queries[i];
  PairTopK<K> top;
  knn_search<K, 4>(g, sorted, cell_start, q.x, q.y, q.z, -1, k, top, lds_bounds + pair, sub, lane);
  static_for<0, H>([&](auto S) {
    const int slot = sub * H + S.value;
    if (slot < k) {
      const int pos = top.part.template id<S.value>();
      out_idx[(size_t)i * k + slot] = pos >= 0 ? __float_as_int(sorted[pos].w) : -1;
      out_d2[(size_t)i * k + slot] = pos >= 0 ? top.part.template d<S.value>() : __builtin_inff();
    }
  });
}

// ---------------------------------------------------------------------------------------------
// Covariance layout conversion at the API boundary (Eigen::Matrix4d image <-> packed symmetric)
// ---------------------------------------------------------------------------------------------
// packed sorted -> N x 16 column-major in ORIGINAL order
__global__ void __launch_bounds__(256) k_covs_expand(const double* __restrict__ covs6, const int* __restrict__ perm, int n, double* __restrict__ out16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* c = covs6 + (size_t)i * 6;
  double* o = out16 + (size_t)perm[i] * 16;
  o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = 0;
  o[4] = c[1]; o[5] = c[3]; o[6] = c[4]; o[7] = 0;
  o[8] = c[2]; o[9] = c[4]; o[10] = c[5]; o[11] = 0;
  o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 0;
}
// N x 16 column-major in ORIGINAL order -> packed sorted (upper triangle as stored: (r<=c) entries)
__global__ void __launch_bounds__(256) k_covs_pack(const double* __restrict__ in16, const int* __restrict__ perm, int n, double* __restrict__ covs6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* m = in16 + (size_t)perm[i] * 16;
  double* c = covs6 + (size_t)i * 6;
  // column-major: m[col*4+row]
  c[0] = m[0];   // (0,0)
  c[1] = m[4];   // (0,1)
  c[2] = m[8];   // (0,2)
  c[3] = m[5];   // (1,1)
  c[4] = m[9];   // (1,2)
  c[5] = m[10];  // (2,2)
}
// reorder packed covs between two sorted orders of the same cloud: dst[i] = src[inv_src[perm_dst[i]]]
__global__ void __launch_bounds__(256) k_invert_perm(const int* __restrict__ perm, int n, int* __restrict__ inv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[perm[i]] = i;
}
__global__ void __launch_bounds__(256) k_covs_reorder(const double* __restrict__ src6, const int* __restrict__ inv_src, const int* __restrict__ perm_dst, int n,
                                                       double* __restrict__ dst6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* s = src6 + (size_t)inv_src[perm_dst[i]] * 6;
  double* d = dst6 + (size_t)i * 6;
#pragma unroll
  for (int e = 0; e < 6; ++e) d[e] = s[e];
}

}  // namespace ngk
