// The per-iteration hot path: ONE fused kernel per Gauss-Newton/LM trial + one single-block solver.
//
//   k_gicp_pass   (data-parallel over source points, G lanes cooperate on one query)
//       K4  error of the trial pose under the PREVIOUS correspondences / Mahalanobis matrices
//           (NanoGICP::compute_error, /root/reference/include/nano_gicp/impl/nano_gicp_impl.hpp:273-296)
//       K2  float transform, exact 1-NN in the target grid, distance gate, Mahalanobis
//           (NanoGICP::update_correspondences, impl/nano_gicp_impl.hpp:174-211)
//       K3  residual, SE(3) Jacobian, H += J^T M J, b += J^T M e, y0 += e^T M e
//           (NanoGICP::linearize, impl/nano_gicp_impl.hpp:214-270)
//       R0  wavefront-shuffle -> LDS -> per-block partial sums (fixed order: deterministic)
//   k_lm_solve    (one block)
//       final reduction of the block partials, then the Levenberg-Marquardt / Gauss-Newton state
//       machine of LsqRegistration (impl/lsq_registration_impl.hpp:89-208) on one lane: 6x6 LDLT,
//       so3_exp, gain ratio, lambda schedule, convergence test, next trial pose.
//
// Why K4 and K2+K3 are fused: the reference evaluates a trial pose xi with compute_error() and, when
// the trial is accepted (the common case), immediately re-linearises at x0 = xi.  Both passes stream
// the same source points, so one pass evaluates the trial's error under the old correspondences AND
// speculatively linearises at xi into the other half of a ping-pong buffer.  If the solver rejects the
// trial, the speculative half is simply overwritten by the next pass; the state visible to the LM
// logic is exactly the reference's (same y0, yi, rho, lambda sequence).  The host never reads back
// inside the loop: the solver publishes a `done` flag and later launches exit immediately.
#pragma once
#include "ngicp_knn.h"

namespace ngk {

constexpr int kPartialStride = 32;  // doubles per block partial: 21 H + 6 b + y0 + yi (+3 pad)
constexpr int kNumSums = 29;
constexpr int kNumSlots = 31;      // + candidates tested, valid correspondences (exact integers carried as doubles)
constexpr int kTraceCols = 8;

struct LmConfig {
  int max_iterations;
  int lm_max_iterations;
  int optimizer;  // 0 GN, 1 LM
  double rot_eps, trans_eps;
  double lm_init_lambda_factor;
};

// Device-resident optimiser state (one per handle).  The solver keeps a private register copy of the
// `hot` part while it works (one batch of loads, one batch of stores; no store->load round trips
// through memory on its serial critical path); `cold` fields are write-only for the solver.
struct LmHot {
  Pose x0;      // current estimate
  Pose xi;      // trial pose the next pass evaluates (== x0 for the first pass / GN)
  Pose delta;   // last step
  double H[36]; // row-major (symmetric), current linearisation
  double b[6];
  double y0;
  double d[6];
  double lambda, nu;
  double cand_total;   // sum over passes of target points distance-tested
  double valid_total;  // sum over passes of gated-in correspondences
  int iter;       // outer iteration index i (impl/lsq_registration_impl.hpp:101-102)
  int trial;      // LM trial index
  int have_lin;   // 0 until the first linearisation exists
  int cur;        // ping-pong half holding the CURRENT correspondences / Mahalanobis
  int done;
  int converged;
  int nr_iterations;
  int lm_failed;
  int n_trace;
  int passes;     // passes that did work
};
struct LmState {
  LmHot hot;
  float xi_f[12];  // float(xi): rows of [R|t], the matrix used for the NN query (impl/nano_gicp_impl.hpp:178)
  double final_hessian[36];
};

struct PassArgs {
  const float4* src;        // sorted source points
  const double* cov_src;    // [n][6], source sorted order
  int n_src;
  const float4* tgt;        // sorted target points
  const int* tgt_cell_start;
  const double* cov_tgt;    // [n_tgt][6], target sorted order
  Grid grid;                // target grid
  int* corr[2];             // [n_src] sorted target position or -1
  double* mahal[2];         // [n_src][6]
  double gate_sq;           // corr_dist_threshold_^2 (double, impl/nano_gicp_impl.hpp:195)
  float gate_sq_f;          // float upper bound of gate_sq for ring termination
  LmState* st;
  double* partials;         // [kNumSlots][partial_pitch], slot-major
  int partial_pitch;        // >= gridDim.x
  int mode;                 // bit0: error part, bit1: linearise part, bit2: ignore st->done (test hooks)
};

// --- cooperative exact 1-NN ------------------------------------------------------------------
template <int G>
__device__ __forceinline__ void group_min(float& d, int& p) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    const float od = __shfl_xor(d, o);
    const int op = __shfl_xor(p, o);
    const bool take = (od < d) || (od == d && (unsigned)op < (unsigned)p);
    d = take ? od : d;
    p = take ? op : p;
  }
}

// Scan one contiguous run of sorted target points (used by the rare outer shells).  Loads are issued
// four at a time so that four memory latencies overlap; the tail batch re-reads the run's last point
// (a duplicate can never win because the comparison is strict), so no per-load predicate is needed.
__device__ __forceinline__ void scan_run_nn(const float4* __restrict__ tgt, int s, int e, float qx, float qy, float qz, float& best, int& pos) {
  const int last = e - 1;
  for (int p = s; p < e; p += 4) {
    const int p1 = min(p + 1, last), p2 = min(p + 2, last), p3 = min(p + 3, last);
    const float4 c0 = tgt[p], c1 = tgt[p1], c2 = tgt[p2], c3 = tgt[p3];
    const float d0 = sqdist(qx, qy, qz, c0), d1 = sqdist(qx, qy, qz, c1), d2 = sqdist(qx, qy, qz, c2), d3 = sqdist(qx, qy, qz, c3);
    // strict '<': first visited wins among equals (impl/nanoflann_impl.hpp:184-211,1368)
    if (d0 < best) { best = d0; pos = p; }
    if (d1 < best) { best = d1; pos = p1; }
    if (d2 < best) { best = d2; pos = p2; }
    if (d3 < best) { best = d3; pos = p3; }
  }
}

// Scan up to NR runs as ONE virtual list, W loads in flight per step: the latency chain of a lane is
// ceil(total / W) memory round trips, whatever the number of runs.
template <int NR, int W>
__device__ __forceinline__ void scan_runs_merged(const float4* __restrict__ tgt, const int (&rs)[NR], const int (&re)[NR], float qx, float qy, float qz, float& best,
                                                 int& pos) {
  int off[NR + 1];
  off[0] = 0;
#pragma unroll
  for (int k = 0; k < NR; ++k) off[k + 1] = off[k] + (re[k] - rs[k]);
  const int total = off[NR];
  if (total == 0) return;
  for (int f = 0; f < total; f += W) {
    int idx[W];
    float4 c[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int g = min(f + j, total - 1);  // the tail repeats the last candidate (strict '<' ignores it)
      int p = rs[0] + g;
#pragma unroll
      for (int k = 1; k < NR; ++k) p = (g >= off[k]) ? rs[k] + (g - off[k]) : p;
      idx[j] = p;
      c[j] = tgt[p];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const float d = sqdist(qx, qy, qz, c[j]);
      if (d < best) {
        best = d;
        pos = idx[j];
      }
    }
  }
}

// G lanes (sub = 0..G-1) search the nearest target point of q.  All G lanes return the same result.
// Rows (fixed y,z; contiguous in x) are dealt round-robin to the lanes of the group and every lane
// walks its own rows, so the lanes of a group — and the groups of a wave — scan concurrently.
template <int G>
__device__ __forceinline__ void nn_search(const Grid& g, const float4* __restrict__ tgt, const int* __restrict__ cell_start, float qx, float qy, float qz,
                                          float gate_sq_f, int sub, float& best, int& pos, unsigned int& ncand) {
  best = 3.4028234664e38f;
  pos = -1;
  int cx, cy, cz;
  cell_coords(g, qx, qy, qz, cx, cy, cz);
  const int rmax = max(max(g.nx, g.ny), g.nz);
  // ring 0 and ring 1 together: 9 rows of up to 3 contiguous cells
  {
    constexpr int NR = (9 + G - 1) / G;
    const int xa = max(cx - 1, 0), xb = min(cx + 1, g.nx - 1);
    int rs[NR], re[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {  // all row bounds first: the cell_start latencies overlap
      const int t = sub + k * G;
      const int z = cz + t / 3 - 1, y = cy + t % 3 - 1;
      const bool ok = (t < 9) && z >= 0 && z < g.nz && y >= 0 && y < g.ny;
      const int row = ok ? (z * g.ny + y) * g.nx : 0;
      const int s = cell_start[row + xa], e = cell_start[row + xb + 1];
      rs[k] = ok ? s : 0;
      re[k] = ok ? e : 0;
      ncand += (unsigned)(re[k] - rs[k]);
    }
    scan_runs_merged<NR, 8>(tgt, rs, re, qx, qy, qz, best, pos);
  }
  if (G > 1) group_min<G>(best, pos);
  for (int r = 1;; ++r) {
    const float bound = unexplored_bound_sq(g, qx, qy, qz, cx, cy, cz, r);
    if (best <= bound || bound >= gate_sq_f || r >= rmax) break;
    // shell r + 1: rows t = sub, sub + G, ... of the (2R+1)^2 (y,z) window clipped to the grid
    const int R = r + 1;
    const int z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
    const int xa = max(cx - R, 0), xb = min(cx + R, g.nx - 1);
    const int wy = y1 - y0 + 1;
    const int nrows = wy * (z1 - z0 + 1);
    for (int t = sub; t < nrows; t += G) {
      const int z = z0 + t / wy, y = y0 + t % wy;
      const int row = (z * g.ny + y) * g.nx;
      if (z == cz - R || z == cz + R || y == cy - R || y == cy + R) {
        const int s = cell_start[row + xa], e = cell_start[row + xb + 1];
        ncand += (unsigned)(e - s);
        scan_run_nn(tgt, s, e, qx, qy, qz, best, pos);
      } else {
        const bool lo = cx - R >= 0, hi = cx + R <= g.nx - 1;
        const int s0 = lo ? cell_start[row + cx - R] : 0, e0 = lo ? cell_start[row + cx - R + 1] : 0;
        const int s1 = hi ? cell_start[row + cx + R] : 0, e1 = hi ? cell_start[row + cx + R + 1] : 0;
        ncand += (unsigned)(e0 - s0) + (unsigned)(e1 - s1);
        scan_run_nn(tgt, s0, e0, qx, qy, qz, best, pos);
        scan_run_nn(tgt, s1, e1, qx, qy, qz, best, pos);
      }
    }
    if (G > 1) group_min<G>(best, pos);
  }
}

// --- reductions --------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// --- the fused pass ------------------------------------------------------------------------------
// Work decomposition: a wave owns batches of B consecutive (cell-sorted) source points.
//   phase 1  search: 64/G groups of G lanes each search one query; B*G/64 rounds cover the batch;
//            the winning (distance, position) of query l is handed to lane l by a wave shuffle;
//   phase 2  FP64 tail on lanes 0..B-1, one query per lane: K4 error under the previous
//            correspondences, gate, Mahalanobis, residual/Jacobian/normal equations.
// Waves never synchronise with each other inside the loop; the only barrier is the final
// block-level reduction.  partials are stored slot-major ([slot][block]) so that the solver's
// reduction reads them coalesced.
template <int G>
__global__ void __launch_bounds__(256) k_gicp_pass(PassArgs a) {
  constexpr int GROUPS = 64 / G;
  constexpr int B = GROUPS > 16 ? GROUPS : 16;
  constexpr int ROUNDS = B / GROUPS;
  __shared__ double lds[4][kNumSlots];
  const LmState* __restrict__ st = a.st;
  if (!(a.mode & 4) && st->hot.done) return;

  const bool do_err = (a.mode & 1) && st->hot.have_lin;
  const bool do_lin = (a.mode & 2);
  const int cur = st->hot.cur, nxt = cur ^ 1;
  const int* __restrict__ corr_old = a.corr[cur];
  const double* __restrict__ mahal_old = a.mahal[cur];
  int* __restrict__ corr_new = a.corr[nxt];
  double* __restrict__ mahal_new = a.mahal[nxt];

  // trial pose (FP64) and its float cast
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = st->hot.xi.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = st->hot.xi.t[i];
  float Tf[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) Tf[i] = st->xi_f[i];

  double acc[kNumSums];
#pragma unroll
  for (int i = 0; i < kNumSums; ++i) acc[i] = 0.0;
  unsigned int ncand = 0, nvalid = 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G, grp = lane / G;
  const int nbatches = (a.n_src + B - 1) / B;
  for (int batch = blockIdx.x * 4 + wave; batch < nbatches; batch += gridDim.x * 4) {
    const int qbase = batch * B;
    float mybest = 3.4028234664e38f;
    int mypos = -1;
    // operands of the FP64 tail that do not depend on the search are requested first, so that their
    // latency hides behind phase 1
    const int i = qbase + lane;
    const bool mine = lane < B && i < a.n_src;
    float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
    int j_old = -1;
    double Mold[6] = {0, 0, 0, 0, 0, 0}, ca[6] = {0, 0, 0, 0, 0, 0};
    if (mine) {
      sp = a.src[i];
      if (do_err) {
        j_old = corr_old[i];
        const double* M = mahal_old + (size_t)i * 6;
#pragma unroll
        for (int e = 0; e < 6; ++e) Mold[e] = M[e];
      }
      if (do_lin) {
        const double* CA = a.cov_src + (size_t)i * 6;
#pragma unroll
        for (int e = 0; e < 6; ++e) ca[e] = CA[e];
      }
    }
    float4 bp_old = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mine && j_old >= 0) bp_old = a.tgt[j_old];
    if (do_lin && (a.mode & 8)) {  // DEBUG timing build: fake search result
      mypos = (qbase + lane) % 1000;
      mybest = 0.01f;
    } else if (do_lin) {
      // ---- phase 1: K2 search (impl/nano_gicp_impl.hpp:178,190-192) ----
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int qi = qbase + r * GROUPS + grp;
        float best = 3.4028234664e38f;
        int pos = -1;
        if (qi < a.n_src) {
          const float4 qp = a.src[qi];
          // Eigen 4x4 * 4-vector in float: ((c0*x + c1*y) + c2*z) + c3*1
          const float qx = ((Tf[0] * qp.x + Tf[1] * qp.y) + Tf[2] * qp.z) + Tf[3];
          const float qy = ((Tf[4] * qp.x + Tf[5] * qp.y) + Tf[6] * qp.z) + Tf[7];
          const float qz = ((Tf[8] * qp.x + Tf[9] * qp.y) + Tf[10] * qp.z) + Tf[11];
          nn_search<G>(a.grid, a.tgt, a.tgt_cell_start, qx, qy, qz, a.gate_sq_f, sub, best, pos, ncand);
        }
        // hand query (r*GROUPS + g)'s result to lane r*GROUPS + g
        const int src_lane = (lane % GROUPS) * G;
        const float gb = __shfl(best, src_lane);
        const int gp = __shfl(pos, src_lane);
        if (lane / GROUPS == r) {
          mybest = gb;
          mypos = gp;
        }
      }
    }
    // ---- phase 2: one query per lane ----
    if ((a.mode & 16) && mine) {  // DEBUG timing build: skip the FP64 tail
      corr_new[i] = mypos;
      acc[27] += (double)mybest;
    } else if (mine) {
      const double ax = (double)sp.x, ay = (double)sp.y, az = (double)sp.z;
      // T * a in FP64 (impl/nano_gicp_impl.hpp:238,289)
      const double tax = R[0] * ax + R[1] * ay + R[2] * az + t[0];
      const double tay = R[3] * ax + R[4] * ay + R[5] * az + t[1];
      const double taz = R[6] * ax + R[7] * ay + R[8] * az + t[2];

      // K4: error of the trial pose under the previous correspondences (impl/nano_gicp_impl.hpp:273-296)
      if (do_err && j_old >= 0) {
        const double ex = (double)bp_old.x - tax, ey = (double)bp_old.y - tay, ez = (double)bp_old.z - taz;
        const double m00 = Mold[0], m01 = Mold[1], m02 = Mold[2], m11 = Mold[3], m12 = Mold[4], m22 = Mold[5];
        const double mex = m00 * ex + m01 * ey + m02 * ez;
        const double mey = m01 * ex + m11 * ey + m12 * ez;
        const double mez = m02 * ex + m12 * ey + m22 * ez;
        acc[28] += ex * mex + ey * mey + ez * mez;
      }
      if (do_lin) {
        const int pos = mypos;
        const bool valid = (pos >= 0) && ((double)mybest < a.gate_sq);  // impl/nano_gicp_impl.hpp:195
        corr_new[i] = valid ? pos : -1;
        if (valid) {
          ++nvalid;
          // Mahalanobis: (C_B + R C_A R^T)^-1  (impl/nano_gicp_impl.hpp:205-209)
          const double* CB = a.cov_tgt + (size_t)pos * 6;
          const float4 bp = a.tgt[pos];
          double rcr[6], M[6];
          rotate_sym(R, ca, rcr);
#pragma unroll
          for (int e = 0; e < 6; ++e) rcr[e] = CB[e] + rcr[e];
          inv3_sym(rcr, M);
          double* Mo = mahal_new + (size_t)i * 6;
#pragma unroll
          for (int e = 0; e < 6; ++e) Mo[e] = M[e];

          // K3: residual, Jacobian, normal equations (impl/nano_gicp_impl.hpp:232-257)
          const double ex = (double)bp.x - tax, ey = (double)bp.y - tay, ez = (double)bp.z - taz;
          const double m00 = M[0], m01 = M[1], m02 = M[2], m11 = M[3], m12 = M[4], m22 = M[5];
          const double mex = m00 * ex + m01 * ey + m02 * ez;
          const double mey = m01 * ex + m11 * ey + m12 * ez;
          const double mez = m02 * ex + m12 * ey + m22 * ez;
          acc[27] += ex * mex + ey * mey + ez * mez;
          // J = [S | -I], S = skew(Ta).   A = S*M  (column j of A = Ta x M[:,j]) = H_rot,trans block
          const double A00 = tay * m02 - taz * m01, A10 = taz * m00 - tax * m02, A20 = tax * m01 - tay * m00;
          const double A01 = tay * m12 - taz * m11, A11 = taz * m01 - tax * m12, A21 = tax * m11 - tay * m01;
          const double A02 = tay * m22 - taz * m12, A12 = taz * m02 - tax * m22, A22 = tax * m12 - tay * m02;
          // H_rr = S^T M S = -(A S);  S columns: (0,az,-ay) (-az,0,ax) (ay,-ax,0)
          acc[0] += -(A01 * taz - A02 * tay);   // (0,0)
          acc[1] += -(-A00 * taz + A02 * tax);  // (0,1)
          acc[2] += -(A00 * tay - A01 * tax);   // (0,2)
          acc[6] += -(-A10 * taz + A12 * tax);  // (1,1)
          acc[7] += -(A10 * tay - A11 * tax);   // (1,2)
          acc[11] += -(A20 * tay - A21 * tax);  // (2,2)
          // H_rt = -S^T M = S M = A   rows 0..2, cols 3..5
          acc[3] += A00; acc[4] += A01; acc[5] += A02;
          acc[8] += A10; acc[9] += A11; acc[10] += A12;
          acc[12] += A20; acc[13] += A21; acc[14] += A22;
          // H_tt = M
          acc[15] += m00; acc[16] += m01; acc[17] += m02;
          acc[18] += m11; acc[19] += m12;
          acc[20] += m22;
          // b = J^T M e = [ S^T Me ; -Me ],  S^T v = v x Ta
          acc[21] += mey * taz - mez * tay;
          acc[22] += mez * tax - mex * taz;
          acc[23] += mex * tay - mey * tax;
          acc[24] += -mex;
          acc[25] += -mey;
          acc[26] += -mez;
        }
      }
    }
  }

  // ---- R0: wave shuffle -> LDS -> block partial (slot-major) ----
#pragma unroll
  for (int v = 0; v < kNumSums; ++v) {
    const double s = wave_sum(acc[v]);
    if (lane == 0) lds[wave][v] = s;
  }
  unsigned int c = ncand, nv = nvalid;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c += __shfl_xor(c, o);
    nv += __shfl_xor(nv, o);
  }
  if (lane == 0) {
    lds[wave][29] = (double)c;
    lds[wave][30] = (double)nv;
  }
  __syncthreads();
  if (threadIdx.x < kNumSlots) {
    const int v = threadIdx.x;
    a.partials[(size_t)v * a.partial_pitch + blockIdx.x] = ((lds[0][v] + lds[1][v]) + lds[2][v]) + lds[3][v];
  }
}

// Upper-triangular packing used by the pass: index of (r,c), r <= c, in the 21-vector
__host__ __device__ __forceinline__ int tri21(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

// --- the solver ----------------------------------------------------------------------------------
constexpr int kSolveThreads = 256;  // 4 waves: keeps the full VGPR budget for the serial lane

struct SolveArgs {
  LmState* st;
  LmConfig cfg;
  const double* partials;  // [kNumSlots][pitch] slot-major (pitch == 1: one pre-reduced vector)
  int nblocks;
  int pitch;
  double* trace;           // [max_rows][8]
  int max_trace_rows;
  int mode;                // 0: LM/GN state machine; 1: reduce -> H/b/y0 (linearize hook); 2: reduce -> y0 = yi (error hook); 3: reduce only
  double* sums_out;        // optional [kPartialStride] reduced sums (29 sums + 2 counters + 1 pad)
};

__device__ __forceinline__ bool is_converged_dev(const Pose& d, double rot_eps, double trans_eps) {  // impl/lsq_registration_impl.hpp:118-127
  double rmax = 0.0, tmax = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) rmax = fmax(rmax, 1.0 / rot_eps * fabs(d.R[r * 3 + c] - (r == c ? 1.0 : 0.0)));
    tmax = fmax(tmax, 1.0 / trans_eps * fabs(d.t[r]));
  }
  return fmax(rmax, tmax) < 1;
}

// d = (H + lambda I)^-1 (-b); delta = (so3_exp(d[0:3]), d[3:6]); xi = delta * x0
__device__ __forceinline__ void make_trial(LmHot& L, double lambda_add) {
  double A[36], rhs[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = L.H[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    A[i * 6 + i] += lambda_add;
    rhs[i] = -L.b[i];
  }
  ldlt6_solve(A, rhs, L.d);
  pose_identity(L.delta);
  so3_exp_matrix(L.d, L.delta.R);
  L.delta.t[0] = L.d[3];
  L.delta.t[1] = L.d[4];
  L.delta.t[2] = L.d[5];
  pose_mul(L.delta, L.x0, L.xi);
}

// new linearisation (reduced sums in LDS) becomes current
__device__ __forceinline__ void adopt_new(LmHot& L, const double* sums) {
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) L.H[r * 6 + c] = L.H[c * 6 + r] = sums[tri21(r, c)];
#pragma unroll
  for (int i = 0; i < 6; ++i) L.b[i] = sums[21 + i];
  L.y0 = sums[27];
  L.cur ^= 1;
  L.have_lin = 1;
}

// One step of LsqRegistration's optimiser on the register-resident state.  Returns true when H was
// accepted as final_hessian_ (impl/lsq_registration_impl.hpp:155,203).
__device__ __forceinline__ bool lm_advance(LmHot& L, const LmConfig& cfg, const double* sums, double* trace, int max_trace_rows) {
  const double yi = sums[28];
  L.passes += 1;
  L.cand_total += sums[29];
  L.valid_total += sums[30];

  if (cfg.optimizer == 0) {
    // ---- Gauss-Newton: impl/lsq_registration_impl.hpp:142-158, one pass per outer iteration ----
    adopt_new(L, sums);
    L.nr_iterations = L.iter;
    make_trial(L, 0.0);
    L.x0 = L.xi;
    L.converged = is_converged_dev(L.delta, cfg.rot_eps, cfg.trans_eps) ? 1 : 0;
    L.iter += 1;
    if (L.converged || L.iter >= cfg.max_iterations) L.done = 1;
    return true;
  }

  // ---- Levenberg-Marquardt: impl/lsq_registration_impl.hpp:161-208 ----
  if (!L.have_lin) {
    // first pass: linearisation at the initial guess
    adopt_new(L, sums);
    L.nr_iterations = 0;
    if (L.lambda < 0.0) {
      double m = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) m = fmax(m, fabs(L.H[i * 6 + i]));
      L.lambda = cfg.lm_init_lambda_factor * m;
    }
    L.nu = 2.0;
    L.trial = 0;
    if (cfg.lm_max_iterations <= 0) {  // the reference's inner loop would not run: "lm not converged"
      L.lm_failed = 1;
      L.done = 1;
      return false;
    }
    make_trial(L, L.lambda);
    return false;
  }

  double den = 0.0, dn = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    den += L.d[i] * (L.lambda * L.d[i] - L.b[i]);
    dn += L.d[i] * L.d[i];
  }
  const double rho = (L.y0 - yi) / den;
  const bool rejected = rho < 0;  // NaN is accepted, as upstream
  if (trace && L.n_trace < max_trace_rows) {
    double* row = trace + (size_t)L.n_trace * kTraceCols;
    row[0] = L.iter; row[1] = L.trial; row[2] = L.y0; row[3] = yi;
    row[4] = rho; row[5] = L.lambda; row[6] = sqrt(dn); row[7] = rejected ? 0.0 : 1.0;
    L.n_trace += 1;
  }
  if (rejected) {
    if (is_converged_dev(L.delta, cfg.rot_eps, cfg.trans_eps)) {  // :191-194 — x0 stays, step reports success
      L.converged = 1;
      L.done = 1;
      return false;
    }
    L.lambda = L.nu * L.lambda;
    L.nu = 2 * L.nu;
    L.trial += 1;
    if (L.trial >= cfg.lm_max_iterations) {  // :207 -> "lm not converged!!", break (:105-108)
      L.lm_failed = 1;
      L.converged = 0;
      L.done = 1;
      return false;
    }
    make_trial(L, L.lambda);
    return false;
  }
  // accepted (:201-204); final_hessian_ = H is written by the caller BEFORE the state is advanced
  return true;
}

__global__ void __launch_bounds__(kSolveThreads) k_lm_solve(SolveArgs a) {
  __shared__ double wsum[kSolveThreads / 64][kPartialStride];
  __shared__ double sums[kPartialStride];
  LmState* st = a.st;
  if (a.mode == 0 && st->hot.done) return;

  // ---- deterministic reduction of the block partials; all 31 loads of a step are in flight together ----
  double acc[kNumSlots];
#pragma unroll
  for (int v = 0; v < kNumSlots; ++v) acc[v] = 0.0;
  for (int b = threadIdx.x; b < a.nblocks; b += kSolveThreads) {
#pragma unroll
    for (int v = 0; v < kNumSlots; ++v) acc[v] += a.partials[(size_t)v * a.pitch + b];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < kNumSlots; ++v) {
    const double s = wave_sum(acc[v]);
    if (lane == 0) wsum[wave][v] = s;
  }
  __syncthreads();
  if (threadIdx.x < kPartialStride) {
    const int v = threadIdx.x;
    sums[v] = v < kNumSlots ? ((wsum[0][v] + wsum[1][v]) + wsum[2][v]) + wsum[3][v] : 0.0;
  }
  __syncthreads();
  if (a.sums_out && threadIdx.x < kPartialStride) a.sums_out[threadIdx.x] = sums[threadIdx.x];
  if (a.mode == 3) return;  // reduce only (point-sharded stepping: the caller all-reduces sums_out)
  if (threadIdx.x != 0) return;

  LmHot L = st->hot;  // private register copy
  if (a.mode == 1) {  // linearize hook
    adopt_new(L, sums);
    st->hot = L;
    return;
  }
  if (a.mode == 2) {  // compute_error hook
    st->hot.y0 = sums[28];
    return;
  }

  const LmConfig& cfg = a.cfg;
  const bool gn = cfg.optimizer == 0;
  const bool accepted = lm_advance(L, cfg, sums, a.trace, a.max_trace_rows);
  if (accepted) {
#pragma unroll
    for (int i = 0; i < 36; ++i) st->final_hessian[i] = L.H[i];
    if (!gn) {
      // LM accept: x0 = xi, lambda update, convergence, next outer iteration (impl/lsq_registration_impl.hpp:201-204,110)
      const double den_dummy = 0.0;
      (void)den_dummy;
      double den = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) den += L.d[i] * (L.lambda * L.d[i] - L.b[i]);
      const double rho = (L.y0 - sums[28]) / den;
      L.x0 = L.xi;
      const double q = 2 * rho - 1;
      L.lambda = L.lambda * fmax(1.0 / 3.0, 1 - q * q * q);
      L.converged = is_converged_dev(L.delta, cfg.rot_eps, cfg.trans_eps) ? 1 : 0;
      L.iter += 1;
      if (L.converged || L.iter >= cfg.max_iterations) {
        L.done = 1;
      } else {
        // the speculative linearisation at xi (== new x0) becomes current
        adopt_new(L, sums);
        L.nr_iterations = L.iter;
        L.nu = 2.0;
        L.trial = 0;
        make_trial(L, L.lambda);
      }
    }
  }
  if (gn && !L.done) L.xi = L.x0;  // GN: the next pass linearises at the updated estimate
  if (!L.done) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) st->xi_f[r * 4 + c] = (float)L.xi.R[r * 3 + c];
      st->xi_f[r * 4 + 3] = (float)L.xi.t[r];
    }
  }
  st->hot = L;
}

// K5: output cloud = float(T) * source, in ORIGINAL order (pcl::transformPointCloud, impl/lsq_registration_impl.hpp:114)
__global__ void __launch_bounds__(256) k_transform_out(const float4* __restrict__ src_sorted, int n, const float* __restrict__ T_colmajor, float* __restrict__ out_xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = src_sorted[i];
  const int o = __float_as_int(p.w);
  const float* m = T_colmajor;
  // pcl::transformPoint: x*m00 + y*m01 + z*m02 + m03 (float)
  out_xyz[(size_t)o * 3 + 0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
  out_xyz[(size_t)o * 3 + 1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
  out_xyz[(size_t)o * 3 + 2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
}

// map correspondences (sorted source slot -> sorted target position) back to ORIGINAL indices
__global__ void __launch_bounds__(256) k_corr_to_original(const int* __restrict__ corr, const float4* __restrict__ src_sorted, const float4* __restrict__ tgt_sorted,
                                                           int n, int* __restrict__ out_corr, float* __restrict__ out_sqd, const float* __restrict__ xi_f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 sp = src_sorted[i];
  const int o = __float_as_int(sp.w);
  const int j = corr[i];
  out_corr[o] = j >= 0 ? __float_as_int(tgt_sorted[j].w) : -1;
  if (out_sqd) {
    float d = __builtin_inff();
    if (j >= 0) {
      const float qx = ((xi_f[0] * sp.x + xi_f[1] * sp.y) + xi_f[2] * sp.z) + xi_f[3];
      const float qy = ((xi_f[4] * sp.x + xi_f[5] * sp.y) + xi_f[6] * sp.z) + xi_f[7];
      const float qz = ((xi_f[8] * sp.x + xi_f[9] * sp.y) + xi_f[10] * sp.z) + xi_f[11];
      d = sqdist(qx, qy, qz, tgt_sorted[j]);
    }
    out_sqd[o] = d;
  }
}

}  // namespace ngk
