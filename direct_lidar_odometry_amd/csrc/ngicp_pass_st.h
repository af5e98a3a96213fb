// k_gicp_pass_st — the fused per-iteration kernel with a STAGED search (round 3).
//
// Same contract as k_gicp_pass (K4 + K2 + K3 + R0 of /root/reference/include/nano_gicp/impl/nano_gicp_impl.hpp:174-296, see
// ngicp_pass.h), same decomposition (one wave per tile-aligned batch of <= 32 queries, 2 lanes per query, one block per group of
// four batches), same prologue, FP64 tail and reduction.  What differs is how the exact 1-NN (impl/nano_gicp_impl.hpp:187-210) is found.
//
// Round 2's search walked the x-sorted rows of the target grid straight from global memory, one lane per (query, row) unit,
// 16-point windows per dependent step: ~35 scattered gather instructions per wave (64 cache lines each), a chain of 2-4 dependent
// round trips, and the CU's address path as the serial resource (profiles/r02_c3_pass_counters.json: SQ_WAIT_ANY 54 %, a dependent
// window step ~6 k cycles once twelve waves queue behind each other).  Here the wave first copies what its queries can possibly
// need into LDS with a handful of COALESCED LDS-DMA instructions, and the per-(query, row) walks then run against LDS:
//
//   reach    every query has an upper bound on its nearest-neighbour distance before the search starts (the previous correspondence,
//            re-measured at the trial pose; else the distance gate).  Its reach box [q - r, q + r] in cells, clamped to its own cell
//            +- G, is what it may have to look at.  The wave's REGION is the union box of its queries' reach boxes.
//   rows     a (y,z) row of the region is one contiguous x-sorted run of the cell-sorted target.  Queries mark the rows whose (y,z)
//            gap can still beat their bound in an LDS bit mask; the marked rows' bounds are fetched with ONE gather (two dwords per
//            row, a lane per row); a prefix sum over the run lengths lays the runs end to end in a virtual array.
//   stage    the virtual array is cut into chunks of kCap points; a chunk is copied with kCap / 64 global_load_lds_dwordx4 (lane =
//            slot, consecutive lanes read consecutive points of a run: a few cache lines per instruction instead of 64).  The points
//            carry their sorted position in w, so anything found in LDS is a genuine (point, position) pair whatever piece it
//            belongs to: walk windows may overhang a piece into its neighbour, into stale slots of an earlier chunk, or into the
//            sentinels that frame the stage - all of them legitimate candidates or infinitely far.
//   units    (query, row piece) pairs that the gap test and the piece's x-extent do not rule out are compacted into an LDS list
//            with ballots (no atomics) and dealt to the lanes round robin.  A unit is a binary search for qx in the piece (LDS)
//            and 4-point windows to the right and to the left while |dx|^2 + gap can still beat the bound; results meet in the
//            per-query 64-bit LDS atomic min on (distance bits, position): the same total order as before, so correspondences
//            stay bit-equal whatever the exploration order.
//   rounds   queries without a bound (no previous correspondence: every query of an alignment's first pass, the gated-out ones
//            later) first look at ring 1 of their own cell only; those that are not provably exact after it take part in a second
//            round with their (now tight) reach.  Queries whose reach was clamped (+- G cells; G covers the distance gate when
//            there is one) continue with the per-query shell walk, which ends at once for everybody else.
#pragma once

namespace ngk {

template <int WPS> struct StCfg;
template <> struct StCfg<3> { static constexpr int kCap = 384, kRows = 256, kUnits = 320, kSubs = 256; };
template <> struct StCfg<4> { static constexpr int kCap = 256, kRows = 192, kUnits = 320, kSubs = 192; };

constexpr int kStPad = 8;      // sentinel slots in front of and behind the stage (blocks of eight points may overhang a piece by seven)
constexpr int kStGrowMax = 3;  // cells a reach box may extend beyond the query's own cell

template <int CAP, int ROWS, int UNITS, int SUBS>
struct WaveStageSt {
  float4 pts[kStPad + CAP + kStPad];  // staged target points {x, y, z, bitcast(sorted position)}
#ifdef NGICP_ST_QFIRST
  unsigned long long qkey[32];  // per query: nearest so far (distance bits << 32 | position)
  float4 qtab[32];              // per query: transformed coordinates, w = cy | cz << 16
#endif
  union {
    struct {
      // the block of 64 region rows at work: points of the rows before it (the runs laid end to end), points, first x-cell, and where
      // its first seven cells start inside the run (slots counted from the run's first)
      int rw_c[64], rw_n[64], rw_xl[64];
      unsigned short rw_b[64][8];
      unsigned short unit_q[UNITS];  // query | region row << 5
      union {
        struct {
          int sub_a[SUBS];    // query | first slot << 5: eight consecutive slots of one unit
          float sub_g[SUBS];  // (y,z) gap of the unit
        };
        struct {
          int row_xlo[ROWS], row_xhi[ROWS];  // per region row: the x-cells some query can still use (only while the rows are marked)
        };
      };
    };
    double red[16 * 30];  // the per-batch reduction reuses the tables
  };
#ifndef NGICP_ST_QFIRST
  unsigned long long qkey[32];  // per query: nearest so far (distance bits << 32 | position)
  float4 qtab[32];              // per query: transformed coordinates, w = cy | cz << 16
#endif
};

// wave-wide scan / reductions on the DPP path (row shifts inside rows of 16 lanes, then the two row broadcasts): a handful of cycles
// per step, where a shuffle through the LDS crossbar (__shfl_up / __shfl_xor) costs a hundred of dependent latency each
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_incl_scan_add(int v) {
  v += dpp_or<0x111, 0xf>(0, v);  // row_shr:1
  v += dpp_or<0x112, 0xf>(0, v);  // row_shr:2
  v += dpp_or<0x114, 0xf>(0, v);  // row_shr:4
  v += dpp_or<0x118, 0xf>(0, v);  // row_shr:8
  v += dpp_or<0x142, 0xa>(0, v);  // row_bcast:15 -> rows 1 and 3
  v += dpp_or<0x143, 0xc>(0, v);  // row_bcast:31 -> rows 2 and 3
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {  // (the running minimum ends up in lane 63)
  const int id = 0x7fffffff;
  v = min(v, dpp_or<0x111, 0xf>(id, v));
  v = min(v, dpp_or<0x112, 0xf>(id, v));
  v = min(v, dpp_or<0x114, 0xf>(id, v));
  v = min(v, dpp_or<0x118, 0xf>(id, v));
  v = min(v, dpp_or<0x142, 0xa>(id, v));
  v = min(v, dpp_or<0x143, 0xc>(id, v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) {
  const int id = (int)0x80000000;
  v = max(v, dpp_or<0x111, 0xf>(id, v));
  v = max(v, dpp_or<0x112, 0xf>(id, v));
  v = max(v, dpp_or<0x114, 0xf>(id, v));
  v = max(v, dpp_or<0x118, 0xf>(id, v));
  v = max(v, dpp_or<0x142, 0xa>(id, v));
  v = max(v, dpp_or<0x143, 0xc>(id, v));
  return __builtin_amdgcn_readlane(v, 63);
}

#ifdef NGICP_ST_DIAG
#define NG_DIAG(...) __VA_ARGS__
#else
#define NG_DIAG(...)
#endif
#define NG_MARK() NG_DIAG(dbg_mark = __builtin_amdgcn_s_memtime();)
#define NG_LAP(k) NG_DIAG({ const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg_t[k] += now_ - dbg_mark; dbg_mark = now_; })
#define NG_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (a.dbg_stamps && lane == 0) a.dbg_stamps[(size_t)(blockIdx.x * 4 + wave) * kStampStride + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_gicp_pass_st(PassArgs a) {
  constexpr int G = 2, B = 32;
  constexpr int CAP = StCfg<WPS>::kCap, ROWS = StCfg<WPS>::kRows, UNITS = StCfg<WPS>::kUnits, SUBS = StCfg<WPS>::kSubs;
  constexpr int kSubPerUnit = SUBS / 64;
  constexpr int RJ = ROWS / 64;
  static_assert(ROWS % 64 == 0 && CAP % 64 == 0 && SUBS % 64 == 0 && UNITS >= 288 && ROWS <= 2048 && CAP + 2 * kStPad <= 65536, "table sizes");
  static_assert(B == kBatchQueries, "query batches are built for 32 queries (2 lanes per query)");
  using Stage = WaveStageSt<CAP, ROWS, UNITS, SUBS>;
  static_assert(sizeof(double) * 16 * 30 <= sizeof(int) * 3 * 64 + 2 * 64 * 8 + 2 * UNITS + sizeof(int) * 2 * (SUBS > ROWS ? SUBS : ROWS), "the reduction tile must fit the tables it reuses");
  __shared__ double lds[4][kNumSlots];
  __shared__ Stage stage_all[4];
  const LmState* __restrict__ st = a.st;
  if (!(a.mode & 4) && st->hot.done) return;

  const bool do_err = (a.mode & 1) && st->hot.have_lin;
  const bool do_lin = (a.mode & 2);
  const int cur = st->hot.cur, nxt = cur ^ 1;
  const float4* __restrict__ tpt_old = a.tpt[cur];
  const double* __restrict__ mahal_old = a.mahal[cur];
  float4* __restrict__ tpt_new = a.tpt[nxt];
  double* __restrict__ mahal_new = a.mahal[nxt];
  const Grid& g = a.grid;

  // trial pose (FP64) and its float cast
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = st->hot.xi.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = st->hot.xi.t[i];
  float Tf[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) Tf[i] = st->xi_f[i];

  double wave_total = 0.0;  // lane v (< 29) accumulates slot v of this wave
  unsigned int ncand = 0, nvalid = 0, nstaged = 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G, grp = lane / G;
  const unsigned long long lt = (1ull << lane) - 1ull;
  Stage& S = stage_all[wave];
  NG_STAMP(0);
  const int group = (a.grp_order && a.order_valid && *a.order_valid) ? a.grp_order[blockIdx.x] : (int)blockIdx.x;
  const unsigned long long t_start = a.grp_cost ? __builtin_amdgcn_s_memtime() : 0ull;
  if (a.dbg_span && threadIdx.x == 0) a.dbg_span[(size_t)blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime();
  if (a.t_first && blockIdx.x == 0 && threadIdx.x == 0 && !st->hot.have_lin) *a.t_first = __builtin_amdgcn_s_memrealtime();
  for (int item = group * 4 + wave; item < a.n_batches; item = a.n_batches) {
    const int2 it = a.batches[item];
    const int qbase = it.x, qcount = it.y;
    float mybest = 3.4028234664e38f;
    int mypos = -1;
    const int i = qbase + lane;
    const bool mine = lane < qcount;
    NG_STAMP(1);
    // ---- the operands of this lane's own query (lane l <-> query qbase + l: the tail's mapping), ONE round trip; K4 at once ----
    const bool have_prev = st->hot.have_lin != 0;
    float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
    int j_old = -1;
    float4 bp_old = make_float4(0.f, 0.f, 0.f, 0.f);
    double k4 = 0.0;
    if (mine) {
      double Mold[6] = {0, 0, 0, 0, 0, 0};
      sp = a.qpts[i];
      if (have_prev) {  // K4's correspondence and the search's warm start
        bp_old = tpt_old[i];
        j_old = __float_as_int(bp_old.w);
      }
      if (do_err) {
        const double* M = mahal_old + (size_t)i * 6;
#pragma unroll
        for (int e = 0; e < 6; ++e) Mold[e] = M[e];
      }
      // K4: error of the trial pose under the previous correspondences (impl/nano_gicp_impl.hpp:273-296)
      if (do_err && j_old >= 0) {
        const double ax = (double)sp.x, ay = (double)sp.y, az = (double)sp.z;
        const double tax = R[0] * ax + R[1] * ay + R[2] * az + t[0];  // T * a in FP64 (impl/nano_gicp_impl.hpp:289)
        const double tay = R[3] * ax + R[4] * ay + R[5] * az + t[1];
        const double taz = R[6] * ax + R[7] * ay + R[8] * az + t[2];
        const double ex = (double)bp_old.x - tax, ey = (double)bp_old.y - tay, ez = (double)bp_old.z - taz;
        const double m00 = Mold[0], m01 = Mold[1], m02 = Mold[2], m11 = Mold[3], m12 = Mold[4], m22 = Mold[5];
        const double mex = m00 * ex + m01 * ey + m02 * ez;
        const double mey = m01 * ex + m11 * ey + m12 * ez;
        const double mez = m02 * ex + m12 * ey + m22 * ez;
        k4 = ex * mex + ey * mey + ez * mez;
      }
    }

    if (do_lin) {
      // ---- K2 (impl/nano_gicp_impl.hpp:178,190-192): the query of this lane pair, handed over by the lane that loaded it ----
      const bool qok = grp < qcount;
      const float qpx = __shfl(sp.x, grp), qpy = __shfl(sp.y, grp), qpz = __shfl(sp.z, grp);
      const int jp = __shfl(j_old, grp);
      float4 bpo;
      bpo.x = __shfl(bp_old.x, grp); bpo.y = __shfl(bp_old.y, grp); bpo.z = __shfl(bp_old.z, grp); bpo.w = 0.f;
      float qx = 0.f, qy = 0.f, qz = 0.f;
      int cx = 0, cy = 0, cz = 0;
      if (qok) {
        // Eigen 4x4 * 4-vector in float: ((c0*x + c1*y) + c2*z) + c3*1
        qx = ((Tf[0] * qpx + Tf[1] * qpy) + Tf[2] * qpz) + Tf[3];
        qy = ((Tf[4] * qpx + Tf[5] * qpy) + Tf[6] * qpz) + Tf[7];
        qz = ((Tf[8] * qpx + Tf[9] * qpy) + Tf[10] * qpz) + Tf[11];
        cell_coords(g, qx, qy, qz, cx, cy, cz);
      }
      float best = 3.4028234664e38f;
      int pos = -1;
      // Warm start: the previous correspondence is a genuine target point, so taking it as the first candidate keeps the search
      // exact (the total order decides as before) - and it bounds what the query can possibly need.
      if (qok && jp >= 0) {
        best = sqdist(qx, qy, qz, bpo);
        pos = jp;
      }
      auto pack_key = [](float d, int p) { return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)p; };
      // per-wave tables: every stage slot starts out infinitely far (later chunks may leave genuine points of earlier ones behind)
      {
        const float far = 3.0e38f;
        const float4 sent = make_float4(far, far, far, __int_as_float(-1));
#pragma unroll
        for (int s = lane; s < CAP + 2 * kStPad; s += 64) S.pts[s] = sent;
      }
      if (qok && sub == 0) {
        S.qtab[grp] = make_float4(qx, qy, qz, __int_as_float(cy | (cz << 16)));
        S.qkey[grp] = pack_key(best, pos);
      }
      NG_STAMP(2);
      const bool cold = qok && pos < 0;  // no bound of its own: ring 1 first
      NG_DIAG(unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dbg_mark = 0;)
      NG_DIAG(unsigned int dbg_P = 0, dbg_live = 0, dbg_chunks = 0, dbg_units = 0, dbg_drains = 0, dbg_witer = 0, dbg_rows = 0, dbg_bs = 0;)  // diagnostic build only
      int explored = -1;                 // rings around the own cell that are done for this query (what the shell walk continues from)
      const int grow = a.stage_grow;     // host: cells that cover the distance gate, at most kStGrowMax
      for (int round = 0; round < 2; ++round) {
        // ---- who takes part, and how far it has to look ----
        bool active = qok;
        if (round == 1) {
          const float bound1 = unexplored_bound_sq(g, qx, qy, qz, cx, cy, cz, explored < 0 ? 0 : explored);
          active = cold && explored < grow && !(best <= bound1 || bound1 >= a.gate_sq_f);
        }
        if (!__any(active)) break;  // wave-uniform
        const float lim0 = fminf(best, a.gate_sq_f);
        const int kq = (round == 0 && cold) ? 1 : grow;
        int ly = 0, lz = 0, hy = 0, hz = 0;
        {
          // reach: every target point that can beat the bound lies within r of the query in every axis.  Cell assignment is monotone
          // in the coordinate, so the cells of q - r and q + r frame the cells of all such points.
          const float r = lim0 < 1.0e30f ? sqrtf(lim0) * 1.000001f + g.slack : 1.0e5f;  // (no bound: everything up to the clamp; a value the cell conversion cannot overflow on)
          int dummy;
          cell_coords(g, qx, qy - r, qz - r, dummy, ly, lz);
          cell_coords(g, qx, qy + r, qz + r, dummy, hy, hz);
        }
        int Y0 = 0, Z0 = 0, wy = 1, nrows = 0;
        int kk = kq, ay0 = 0, ay1 = 0, az0 = 0, az1 = 0;
        for (int shrink = 0;; ++shrink) {  // the region's rows must fit the row table: take rings off the clamp until they do
          kk = max(kq - shrink, 0);
          ay0 = max(ly, cy - kk); ay1 = min(hy, cy + kk);
          az0 = max(lz, cz - kk); az1 = min(hz, cz + kk);
          const int big = 0x3fffffff;
          Y0 = wave_min_i(active ? ay0 : big);
          Z0 = wave_min_i(active ? az0 : big);
          const int Y1 = wave_max_i(active ? ay1 : -1), Z1 = wave_max_i(active ? az1 : -1);
          wy = Y1 - Y0 + 1;
          nrows = wy * (Z1 - Z0 + 1);
          if (nrows <= ROWS) break;
          if (shrink >= grow) {  // (wave-uniform) even the queries' own cells span more rows than the table holds: leave this batch to the shell walk
            nrows = 0;
            break;
          }
        }
        if (nrows == 0) break;  // wave-uniform
        // ---- every query marks the rows it can still use, with the cells of the row that lie within its reach; every such
        //      (query, row) pair is a UNIT, listed with ballots ----
        wave_lds_sync();  // (the previous round's sub-units share their space with the marks)
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
          S.row_xlo[lane + 64 * j] = 0x3fffffff;
          S.row_xhi[lane + 64 * j] = -1;
        }
        wave_lds_sync();
        int U = 0;
        bool dropped = false;  // a unit of this lane did not fit the list: its query falls back to the shell walk
        {
          const int ny = active ? ay1 - ay0 + 1 : 1, nr = active ? ny * (az1 - az0 + 1) : 0;
          const int nr_max = wave_max_i(nr);
          for (int kb = 0; kb < nr_max; kb += G) {  // wave-uniform trip count
            const int k2 = kb + sub;
            bool on = k2 < nr;
            int tr = 0;
            if (on) {
              const int y = ay0 + k2 % ny, z = az0 + k2 / ny;
              const float gp = row_gap_sq(g, y, z, cy, cz, qy, qz);
              on = gp <= lim0;
              if (on) {
                // a point of this row that can beat the bound has dx^2 <= bound - (dy^2 + dz^2) <= bound - gap
                const float rr = lim0 < 1.0e30f ? sqrtf(lim0 - gp) * 1.000001f + g.slack : 1.0e5f;
                const int xl = max(clampi((int)floorf((qx - rr - g.ox) * g.inv_h), 0, g.nx - 1), cx - kk);
                const int xh = min(clampi((int)floorf((qx + rr - g.ox) * g.inv_h), 0, g.nx - 1), cx + kk);
                tr = (z - Z0) * wy + (y - Y0);
                atomicMin(&S.row_xlo[tr], xl);
                atomicMax(&S.row_xhi[tr], xh);
              }
            }
            const unsigned long long m = __ballot(on);
            if (on) {
              const int slot = U + __popcll(m & lt);
              if (slot < UNITS) S.unit_q[slot] = (unsigned short)(grp | (tr << 5));
              else dropped = true;
            }
            U = min(U + __popcll(m), UNITS);
          }
        }
        wave_lds_sync();
        if (round == 0) NG_STAMP(3);
        // (the marks are read NOW, for every block of rows: the sub-unit lists of the first block will reuse their space)
        int mxl[RJ], mxh[RJ];
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
          mxl[j] = S.row_xlo[lane + 64 * j];
          mxh[j] = S.row_xhi[lane + 64 * j];
        }
        wave_lds_sync();
        // ---- blocks of 64 region rows (one is the rule) ----
        for (int rb0 = 0; rb0 < nrows; rb0 += 64) {
          // bounds of the marked rows (a lane per row, ONE round trip): the run [s, e) of the cells xl..xh and the starts of its first
          // seven cells (what lies beyond the seventh is one bucket)
          const int tr = rb0 + lane;
          int rs = 0, rn = 0, rxl = 0;
          int bnd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          {
            int xl = mxl[0], xh = mxh[0];
#pragma unroll
            for (int v = 1; v < RJ; ++v)
              if ((rb0 >> 6) == v) xl = mxl[v], xh = mxh[v];
            const bool has = tr < nrows && xh >= xl;
            if (has) {
              const int y = Y0 + tr % wy, z = Z0 + tr / wy;
              const int* cs = a.tgt_cell_start + ((z * g.ny + y) * g.nx + xl);
              struct alignas(4) I4 { int v[4]; };
              const I4 b0 = *reinterpret_cast<const I4*>(cs), b1 = *reinterpret_cast<const I4*>(cs + 4);  // (the table is padded for it)
              const int ev = cs[xh + 1 - xl];
              rs = b0.v[0];
              rn = ev - rs;
              rxl = xl;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                bnd[k] = min(b0.v[k], ev) - rs;
                bnd[4 + k] = min(b1.v[k], ev) - rs;
              }
            }
          }
          const int rinc = wave_incl_scan_add(rn);
          const int rc = rinc - rn;  // points of the block's rows before this one: the rows' runs laid end to end
          const int P = __builtin_amdgcn_readlane(rinc, 63);
          if (lane == 0) nstaged += (unsigned int)P;
          NG_DIAG(dbg_P += (unsigned int)P; dbg_live += (unsigned int)__popcll(__ballot(rn > 0));)
          wave_lds_sync();  // (the previous block's sub-units are done with the row tables)
          S.rw_c[lane] = rc;
          S.rw_n[lane] = rn;
          S.rw_xl[lane] = rxl;
#pragma unroll
          for (int k = 0; k < 8; ++k) S.rw_b[lane][k] = (unsigned short)min(bnd[k], 0xffff);
          for (int c0 = 0; c0 < P; c0 += CAP) {
            const int c1 = min(c0 + CAP, P);
            NG_DIAG(++dbg_chunks;)
            const unsigned long long pieces = __ballot(rn > 0 && rc < c1 && rc + rn > c0);  // rows with a piece in this chunk, in row order
            wave_lds_sync();  // the previous chunk's sub-units are done with the stage
            NG_MARK()
            // ---- copy: one piece after the other, lane = point (consecutive lanes read consecutive points of the run) ----
            for (unsigned long long m = pieces; m; m &= m - 1) {  // wave-uniform
              const int b = __builtin_ctzll(m);
              const int ps = __builtin_amdgcn_readlane(rs, b), pc = __builtin_amdgcn_readlane(rc, b), pn = __builtin_amdgcn_readlane(rn, b);
              const int v0 = max(pc, c0), v1 = min(pc + pn, c1);  // the piece in virtual coordinates
              const float4* src0 = a.tgtp + (ps + (v0 - pc));
              for (int o = 0; o < v1 - v0; o += 64) {
                if (o + lane < v1 - v0)
                  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + o + lane), (__attribute__((address_space(3))) void*)&S.pts[kStPad + (v0 - c0) + o], 16, 0, 0);
              }
            }
            NG_LAP(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_sync();
            NG_LAP(1)
            const float4* __restrict__ PT = S.pts + kStPad;
            // ---- the units of this chunk, a unit per lane: which slots can still hold a better point (whole cells of the row: no
            //      search), cut into blocks of eight consecutive slots ----
            for (int u0 = 0; u0 < U; u0 += 64) {  // wave-uniform
              int qs = 0, sa = 0, nb = 0, nv = 0;  // query, first slot, blocks, slots
              float gyz = 0.f;
              if (u0 + lane < U) {
                const int uq = S.unit_q[u0 + lane], li = (uq >> 5) - rb0;
                qs = uq & 31;
                if (li >= 0 && li < 64) {
                  const int c = S.rw_c[li], n = S.rw_n[li];
                  if (n > 0 && c < c1 && c + n > c0) {
                    const float4 q = S.qtab[qs];
                    const int qcyz = __float_as_int(q.w), trr = uq >> 5;
                    gyz = row_gap_sq(g, Y0 + trr % wy, Z0 + trr / wy, qcyz & 0xffff, qcyz >> 16, q.y, q.z);
                    const float lim = fminf(__uint_as_float((unsigned int)(S.qkey[qs] >> 32)), a.gate_sq_f);
                    if (gyz <= lim) {
                      const float rr = lim < 1.0e30f ? sqrtf(lim - gyz) * 1.000001f + g.slack : 1.0e5f;
                      const int xl = clampi((int)floorf((q.x - rr - g.ox) * g.inv_h), 0, g.nx - 1), xh = clampi((int)floorf((q.x + rr - g.ox) * g.inv_h), 0, g.nx - 1);
                      const int ka = xl - S.rw_xl[li], kb = xh + 1 - S.rw_xl[li];  // cells of the row, counted from its first
                      // (cells in front of the row's first or behind its seventh: the row's ends / the last bucket - a superset)
#ifdef NGICP_ST_NOBUCKET
                      const int ra = 0 * ka;
#else
                      const int ra = ka <= 0 ? 0 : (int)S.rw_b[li][min(ka, 7)];
#endif
                      int rbnd = kb <= 0 ? 0 : (kb > 7 ? n : (int)S.rw_b[li][kb]);
                      if (rbnd == 0xffff) rbnd = n;  // (a saturated table entry: the run's end is the safe answer)
#ifdef NGICP_ST_NOBUCKET
                      rbnd = n;
#endif
                      const int va = max(c + ra, c0), vb = min(c + rbnd, c1);  // virtual coordinates, inside this chunk
                      if (vb > va) {
                        sa = va - c0;
                        nv = vb - va;
                        nb = (nv + 7) >> 3;
                      }
                    }
                  }
                }
              }
              NG_LAP(2)
              // ---- the blocks become sub-units, kSubPerUnit per unit and round, dealt to all lanes ----
              int eb = 0;
              for (;;) {
                const int rem = nb - eb;
                if (!__any(rem > 0)) break;  // wave-uniform
                NG_DIAG(++dbg_witer;)
                const int tk = min(rem, kSubPerUnit);
                const int inc = wave_incl_scan_add(tk);
                const int total = __builtin_amdgcn_readlane(inc, 63);
                int slot = inc - tk;
#pragma unroll
                for (int e = 0; e < kSubPerUnit; ++e) {
                  if (e < tk) {
                    S.sub_a[slot] = qs | (min(8, nv - 8 * eb) << 5) | ((sa + 8 * eb) << 9);  // query | slots that count | first slot
                    S.sub_g[slot] = gyz;
                    ++eb;
                    ++slot;
                  }
                }
                wave_lds_sync();
                NG_LAP(3)
                NG_DIAG(dbg_units += (unsigned int)total; ++dbg_drains;)
                for (int sidx = lane; sidx < total; sidx += 64) {
                  const int rec = S.sub_a[sidx], sq = rec & 31, cnt = (rec >> 5) & 15, start = rec >> 9;
                  const float sg = S.sub_g[sidx];
                  const float4 qq = S.qtab[sq];
                  const unsigned long long k0 = S.qkey[sq];
                  float ub = __uint_as_float((unsigned int)(k0 >> 32));
                  int up = (int)(unsigned int)k0;
                  const float lim = fminf(ub, a.gate_sq_f);
                  const float4 p0 = PT[start], p1 = PT[start + 1], p2 = PT[start + 2], p3 = PT[start + 3], p4 = PT[start + 4], p5 = PT[start + 5], p6 = PT[start + 6], p7 = PT[start + 7];
                  // the unit's slots are sorted by x: a block whose nearest end is out of reach holds nothing for it (the bound may have
                  // moved since the block was cut).  Only the first `cnt` slots belong to the unit; what follows is its neighbour's,
                  // stale, or infinitely far - looked at all the same (genuine points), but not part of the order in x.
                  float xlast = p0.x;
                  xlast = cnt > 1 ? p1.x : xlast; xlast = cnt > 2 ? p2.x : xlast; xlast = cnt > 3 ? p3.x : xlast; xlast = cnt > 4 ? p4.x : xlast;
                  xlast = cnt > 5 ? p5.x : xlast; xlast = cnt > 6 ? p6.x : xlast; xlast = cnt > 7 ? p7.x : xlast;
                  const float dl = p0.x - qq.x, dr = qq.x - xlast;
#ifndef NGICP_ST_NOPRUNE
                  if ((dl > 0.f && dl * dl + sg > lim) || (dr > 0.f && dr * dr + sg > lim)) continue;
#endif
                  auto take = [&](const float4& p) {
                    const float d = sqdist(qq.x, qq.y, qq.z, p);
                    const int pp = __float_as_int(p.w);
                    if (nn_better(d, pp, ub, up)) { ub = d; up = pp; }
                  };
                  take(p0); take(p1); take(p2); take(p3); take(p4); take(p5); take(p6); take(p7);
                  ncand += 8;
                  atomicMin(&S.qkey[sq], pack_key(ub, up));
                }
                wave_lds_sync();
                NG_LAP(4)
              }
            }
          }
        }
        {  // (both lanes of a query must agree: the shuffle is executed by every lane, not behind a short-circuit)
          const int partner = __shfl_xor(dropped ? 1 : 0, 1);
          dropped = dropped || partner != 0;
        }
        wave_lds_sync();
        if (active) {
          const unsigned long long k1 = S.qkey[grp];
          best = __uint_as_float((unsigned int)(k1 >> 32));
          pos = (int)(unsigned int)k1;
          if (!dropped) explored = max(explored, kk);
        }
        if (round == 0) NG_STAMP(4);
      }
      NG_STAMP(5);
      if (a.dbg_qstats && qok && sub == 0) a.dbg_qstats[qbase + grp] = make_int4(pos, __float_as_int(best), explored | (cold ? 256 : 0), __float_as_int(qx));  // diagnostic only
#ifdef NGICP_ST_DIAG
      if (a.dbg_stamps) {  // diagnostic only
        unsigned long long* d = a.dbg_stamps + (size_t)(blockIdx.x * 4 + wave) * kStampStride;
        if (lane == 0) { d[10] = dbg_P; d[11] = dbg_live; d[12] = dbg_chunks; d[13] = dbg_units; d[14] = dbg_drains; d[17] = dbg_rows; d[18] = (unsigned long long)qcount; }
        atomicMax(&d[15], (unsigned long long)dbg_witer);  // the busiest lane's window steps / binary-search probes
        atomicMax(&d[16], (unsigned long long)dbg_bs);
        if (lane == 0) { d[19] = dbg_t[0]; d[20] = dbg_t[1]; d[21] = dbg_t[2]; d[22] = dbg_t[3]; d[23] = dbg_t[4] | (dbg_t[5] << 32); }
      }
#endif
      // whatever lies beyond the clamped reach (an unbounded or very wide gate, a batch that did not fit the tables): per query
      if (qok) nn_shells<G>(g, a.tgtp, a.tgt_cell_start, qx, qy, qz, cx, cy, cz, a.gate_sq_f, sub, explored, best, pos, ncand);
      if (a.dbg_qstats && qok && sub == 0) a.dbg_qstats[qbase + grp] = make_int4(pos, __float_as_int(best), explored | (cold ? 256 : 0) | 512, __float_as_int(qx));  // diagnostic only: after the shell walk
      NG_STAMP(6);
      // hand query g's result to lane g
      const int src_lane = (lane % B) * G;
      mybest = __shfl(best, src_lane);
      mypos = __shfl(pos, src_lane);
    }
    double acc[kNumSums];
#pragma unroll
    for (int v = 0; v < kNumSums; ++v) acc[v] = 0.0;
    if (mine) {
      const double ax = (double)sp.x, ay = (double)sp.y, az = (double)sp.z;
      // T * a in FP64 (impl/nano_gicp_impl.hpp:238,289)
      const double tax = R[0] * ax + R[1] * ay + R[2] * az + t[0];
      const double tay = R[3] * ax + R[4] * ay + R[5] * az + t[1];
      const double taz = R[6] * ax + R[7] * ay + R[8] * az + t[2];

      acc[28] += k4;  // K4, evaluated before the search
      if (do_lin) {
        const int pos = mypos;
        const bool valid = (pos >= 0) && ((double)mybest < a.gate_sq);  // impl/nano_gicp_impl.hpp:195
        if (a.dbg_qstats) a.dbg_qstats[a.n_src + i] = make_int4(pos, __float_as_int(mybest), valid ? 1 : 0, __float_as_int(sp.y));  // diagnostic only
        if (!valid) tpt_new[i] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (valid) {
          ++nvalid;
          // Mahalanobis: (C_B + R C_A R^T)^-1  (impl/nano_gicp_impl.hpp:205-209)
          const double* CB = a.cov_tgt + (size_t)pos * 6;
          const double* CA = a.cov_src + (size_t)__float_as_int(sp.w) * 6;  // same round trip as the target's
          const float4 bp = a.tgtp[pos];
          double ca[6];
#pragma unroll
          for (int e = 0; e < 6; ++e) ca[e] = CA[e];
          tpt_new[i] = make_float4(bp.x, bp.y, bp.z, __int_as_float(pos));
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
    // ---- R0: per-batch reduction through LDS (the tables are idle now): lanes 0..31 hold the tail's sums; sixteen of them at a
    //      time write a row of a [16][30] tile, lane v then adds column v in fixed order (queries 0, 1, ... 31: deterministic) ----
    {
      double* red = S.red;  // [16][30] doubles
      double out = 0.0;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        wave_lds_sync();
        if ((lane >> 4) == half) {
#pragma unroll
          for (int v = 0; v < kNumSums; ++v) red[(lane & 15) * 30 + v] = acc[v];
        }
        wave_lds_sync();
        if (lane < kNumSums)
          for (int l = 0; l < 16; ++l) out += red[l * 30 + lane];
      }
      if (lane < kNumSums) wave_total += out;
    }
  }

  NG_STAMP(7);
  {
    // counters: [3][64] through LDS, lanes 29..31 add their column
    wave_lds_sync();
    unsigned int* cnt = reinterpret_cast<unsigned int*>(S.red);
    cnt[lane] = ncand;
    cnt[64 + lane] = nvalid;
    cnt[128 + lane] = nstaged;
    wave_lds_sync();
    if (lane >= kNumSums && lane < kNumSlots) {
      unsigned int sum = 0;
      for (int l = 0; l < 64; ++l) sum += cnt[(lane - kNumSums) * 64 + l];
      wave_total = (double)sum;
    }
    if (lane < kNumSlots) lds[wave][lane] = wave_total;
  }
  NG_STAMP(8);
  __syncthreads();
  NG_STAMP(9);
  if (threadIdx.x < kNumSlots) {
    const int v = threadIdx.x;
    a.partials[(size_t)group * kNumSlots + v] = ((lds[0][v] + lds[1][v]) + lds[2][v]) + lds[3][v];
  }
  if (a.dbg_span && threadIdx.x == 0) {
    unsigned long long* d = a.dbg_span + (size_t)blockIdx.x * 4;
    d[1] = __builtin_amdgcn_s_memrealtime();
    d[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID, XCC_ID
    d[3] = (unsigned long long)(unsigned int)group;
  }
  if (a.grp_cost && threadIdx.x == 0) a.grp_cost[group] = (int)min((__builtin_amdgcn_s_memtime() - t_start) >> 4, 0x7fffffffull);
}
#undef NG_STAMP
#undef NG_DIAG
#undef NG_MARK
#undef NG_LAP

}  // namespace ngk
