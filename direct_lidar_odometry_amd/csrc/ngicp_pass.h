// The per-iteration hot path: ONE fused kernel per Gauss-Newton/LM trial + one single-block solver.
//
//   k_gicp_pass   (data-parallel over source points, G lanes cooperate on one query)
//       K4  error of the trial pose under the PREVIOUS correspondences / Mahalanobis matrices
//           (NanoGICP::compute_error, /root/reference/include/nano_gicp/impl/nano_gicp_impl.hpp:273-296)
//       K2  float transform, exact 1-NN in the target grid, distance gate, Mahalanobis
//           (NanoGICP::update_correspondences, impl/nano_gicp_impl.hpp:174-211)
//       K3  residual, SE(3) Jacobian, H += J^T M J, b += J^T M e, y0 += e^T M e
//           (NanoGICP::linearize, impl/nano_gicp_impl.hpp:214-270)
//       R0  LDS transpose -> per-group partial sums (fixed order: deterministic, independent of launch order)
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
#include <type_traits>

#include "ngicp_knn.h"

namespace ngk {

constexpr int kPartialStride = 32;  // doubles per block partial: 21 H + 6 b + y0 + yi (+3 pad)
constexpr int kNumSums = 29;
constexpr int kNumSlots = 32;      // + candidates tested, valid correspondences, queries served through the LDS row list (exact integers carried as doubles)
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
  double staged_total; // sum over passes of queries served through the LDS row list of their batch region
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
  int pending;    // k_gicp_head only: the rows of a finished pass wait for the next launch's head to step the optimiser with them
  double final_H[36];  // H of the last accepted step, row-major (final_hessian_, impl/lsq_registration_impl.hpp:155,203); identity until then
  unsigned long long t_first, t_done;  // 100 MHz counter at the start of the alignment's first pass / when the solver set `done`
};
// What a pass needs of the state, as 64 dwords (256 bytes, two cache lines of their own).  The persistent kernel reads the view of pass p
// from entry p of a RING of views that begins at LmState::view and continues behind the state (the solver that ends pass p - 1 stores
// it write-through): an address no CU and no L2 of this launch has touched before cannot be stale anywhere, so the blocks read it
// with plain scalar loads, exactly as a launch-per-pass kernel reads the state - and the compiler may reload the values instead of
// keeping 40 registers alive through the search.
constexpr int kViewXi = 0;     // [24] the trial pose xi: R (9 doubles), t (3 doubles)
constexpr int kViewXiF = 24;   // [12] float(xi), rows of [R|t]
constexpr int kViewDone = 36, kViewCur = 37, kViewHaveLin = 38;
struct LmState {
  LmHot hot;
  float xi_f[12];  // float(xi): rows of [R|t], the matrix used for the NN query (impl/nano_gicp_impl.hpp:178)
  // entry 0 of the persistent kernel's ring of per-pass views (kView*); entries 1 .. max_passes follow at a stride of kViewWords.
  // Every entry lies in cache lines of its own (L2: 128 bytes, scalar cache: 64) with an unused gap behind it: reading entry p must
  // not bring any byte of entry p + 1 into a cache before the solver has written it.
  alignas(512) int view[128];
};
constexpr int kViewWords = 128;
// The release word of the persistent kernel exists kGenLines times, 128 bytes apart; block b waits on copy b % kGenLines.  (With ONE
// word, the ~770 waiting blocks' agent-scope polls - every one goes out to memory - queued on a single channel: whatever else mapped
// to that channel waited behind them, and the blocks still working took 2.5x as long.)
constexpr int kGenLines = 256, kGenStride = 32;
static_assert(sizeof(LmState) % 512 == 0 && offsetof(LmState, view) % 512 == 0, "ring entries are 512-byte aligned");

struct alignas(4) Xyz { float x, y, z; };
__device__ __forceinline__ float sqdist(float qx, float qy, float qz, const Xyz& p) {
  const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// sorted float4 points (with their sentinel frame) -> the same points carrying their sorted position instead of their original index
__global__ void __launch_bounds__(256) k_pack_pos(const float4* __restrict__ in, int n_padded, int pad, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_padded) return;
  const float4 p = in[i];
  const bool real = i >= pad && i < n_padded - pad;
  out[i] = make_float4(p.x, p.y, p.z, __int_as_float(real ? i - pad : -1));
}

// sorted float4 points (with their sentinel frame) -> the packed copy
__global__ void __launch_bounds__(256) k_pack_xyz(const float4* __restrict__ in, int n, Xyz* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  out[i] = Xyz{p.x, p.y, p.z};
}

struct SolveArgs {
  LmState* st;
  LmConfig cfg;
  const double* partials;  // [nblocks][kNumSlots] group-major (nblocks == 1: one pre-reduced vector)
  int nblocks;
  double* trace;           // [max_rows][8]
  int max_trace_rows;
  int mode;                // 0: LM/GN state machine; 1: reduce -> H/b/y0 (linearize hook); 2: reduce -> y0 = yi (error hook); 3: reduce only
  double* sums_out;        // optional [kPartialStride] reduced sums (29 sums + 2 counters + 1 pad)
  int* grp_order;          // [nblocks] out: groups sorted by measured cost, heaviest first (mode 0 only), or null
  const int* grp_cost;     // [nblocks] in: duration of each group's block in the pass just finished
  unsigned long long* dbg_stamps;  // diagnostic only: [8] s_memtime stamps of the last launch, or null
  int* progress_host;      // pinned host memory, or null: {passes done | kProgressDone} published after every step (mode 0)
  LmHot* final_host;       // pinned host memory, or null: the state image, written when the alignment is done, BEFORE the done flag goes out
  int* order_valid;        // device word: the solver has published a group order (heaviest first) for the next pass
  const unsigned long long* t_first;  // device word the first pass of an alignment stamps (100 MHz counter)
  LmState* st_out;         // where the advanced state goes (null: back to st).  k_gicp_head: the blocks of the launch read st while its solver block writes
  int nrows;               // rows of `partials` (0: nblocks, one per group).  k_gicp_head: 32 rows, already added up per subset by the pass's blocks
  int persist;             // the solver runs between two passes of ONE launch: every word other blocks (on other XCDs) wrote or will read
                           // goes through agent-scope loads / write-through stores, and the launch order it builds is for the NEXT alignment
  unsigned long long* pass_ticks;  // persist: [2 * max passes] 100 MHz ticks {last block arrived, next pass released} per pass, or null
};

struct PassArgs {
  const float4* qpts;       // source points in Morton-tile query order, w = sorted source position
  const int2* batches;      // tile-aligned query batches {first qpts index, count <= 32}
  const float* batch_boxes; // [n_batches][6] centre + half extents of each batch in the source frame
  const int* grp_order;     // [n_groups] launch order of the groups (a block takes group grp_order[blockIdx.x]); valid when st->order_valid
  int* grp_cost;            // [n_groups] measured duration of each group's block in the last pass (shader clocks >> 4), or null
  int n_batches;
  const double* cov_src;    // [n][6], source sorted order
  int n_src;
  const float4* tgt;        // sorted target points
  const float4* tgtp;       // the same points as {x, y, z, bitcast(sorted position)} (same sentinel frame): what k_gicp_pass_st copies to LDS
  const Xyz* tgt3;          // the same points packed 12 bytes each (same sentinel frame): what the walks and the tail fetch - a quarter
                            // fewer bytes per candidate and three registers per point instead of four
  const int* tgt_cell_start;
  const unsigned int* tgt_cell_box;  // per-cell (y,z) extents of the target's points (k_cell_boxes; framed like the cell-start table), or null
  const double* cov_tgt;    // [n_tgt][6], target sorted order
  Grid grid;                // target grid
  float4* tpt[2];           // [n_src] the correspondence of each query: {x, y, z of the target point, bitcast(sorted target position or -1)}:
                            // the next pass gets the warm start of its search and K4's target point in ONE load (no corr -> point chain)
  double* mahal[2];         // [n_src][6]
  double gate_sq;           // corr_dist_threshold_^2 (double, impl/nano_gicp_impl.hpp:195)
  float gate_sq_f;          // float upper bound of gate_sq for ring termination
  unsigned char* batch_far; // [n_batches] 1 if some query of the batch looked beyond its ring 1 in the previous pass: only then
                            // are the outer rings of the batch region listed (a wrong guess costs time, not exactness)
  LmState* st;
  double* partials;         // [n_groups][kNumSlots], group-major: a block stores its 32 sums as one 256-byte row
  int mode;                 // bit0: error part, bit1: linearise part, bit2: ignore st->done (test hooks), bit5: the first pass lists region rows
  int stage_grow;           // rings around the batch box that the LDS row list covers (0: no list, search unindexed)
  unsigned long long* dbg_stamps;  // diagnostic only: [wave][kStampStride] s_memtime stamps + counters, or null
  int4* dbg_qstats;                // diagnostic only: per query {ring-1 candidates, ring-1 walks | far walks << 16, far + shell candidates, flags}, or null
  unsigned long long* dbg_span;    // diagnostic only: per block {s_memrealtime (10 ns ticks) at entry, at exit, HW_ID | XCC_ID << 32, group}, or null
  // fused solver: the block whose ticket is the last of the grid reduces the group rows and advances the optimiser in the tail of
  // the SAME launch (no second dispatch per iteration).  Rows and costs are then stored write-through and published by the ticket.
  const int* order_valid;          // device word: grp_order holds a complete order
  unsigned long long* t_first;     // device word: stamped by the first pass of an alignment (block 0), or null
  int fused;
  int* ticket;   // zero before the launch; the last block puts it back (persistent: counts on, pass after pass)
  // persistent: ONE launch per alignment.  A block keeps its groups for the whole alignment (their correspondences and Mahalanobis
  // matrices never leave its CU's L1 / its XCD's L2: no kernel boundary drops them), the blocks meet at the ticket after every pass,
  // the last one to arrive steps the optimiser and releases the next pass through `gen`.
  // k_gicp_head: no solver launch.  Every block of launch i steps the optimiser itself, at its head, with the sums of pass i - 1 (32 rows,
  // added up per subset of groups by the blocks that finished their subset last), then searches at the pose it has just computed.
  const double* crow_in;   // [32][kNumSlots] the subset rows of the previous pass (zero rows for subsets without groups)
  double* crow_out;        // [32][kNumSlots] this pass's
  int* cluster_ticket;     // [32] zero before the launch; the last block of a subset puts its word back
  int* done_flag;          // device word, zero at the start of an alignment: a launch's solver block sets it when the alignment is over
  int persist;
  int first_pass;  // (0 unless the kernel is launched once per pass for an A/B measurement: the ring entry the launch begins with)
  int max_passes;
  int* gen;      // zero before the launch: passes released so far (agent scope), kGenLines copies in cache lines of their own
  SolveArgs sa;
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

// Nearest-neighbour candidates are ranked by (squared distance, sorted target position).  The reference
// keeps the first point its kd-tree visits among exactly equidistant ones (impl/nanoflann_impl.hpp:184-211),
// which no other index can mirror (SURVEY.md §7 "Ties"); a total order makes the winner independent of how
// rows are dealt to lanes and of the order in which windows are looked at.
__device__ __forceinline__ bool nn_better(float d, int p, float best, int bestp) { return d < best || (d == best && (unsigned)p < (unsigned)bestp); }

// --- the fused pass ------------------------------------------------------------------------------
// Work decomposition: a wave owns ONE tile-aligned batch of up to 32 consecutive queries in Morton-tile order (a
// compact 3-D blob; batches are cut at index-build time so that none leaves its tile); a block is the four waves of
// four consecutive batches (a GROUP: they share most of their target rows, and with them the CU's L1).  Per batch:
//   index    the wave pushes the batch's bounding box through the trial pose, grows it by as many rings as fit the
//            row table (up to the number that covers the distance gate) and fetches, in ONE round trip, the bounds
//            of every (y,z) row of that region in the cell-sorted target (a row is one contiguous, x-sorted run).
//            The NON-EMPTY rows go into an LDS list, nearest ring first: empty space costs nothing afterwards;
//   search   2 lanes per query set it up (transform, cell, warm start from the previous correspondence).  Every (query, row)
//            pair that the (y,z)-gap test does not rule out becomes a UNIT in an LDS queue; whichever lane is free pops the
//            next unit and walks that row: the bounds of the row's cells around qx, then 12-point windows from where qx sits
//            (interpolated) outward, right / left, only while the x-gap alone does not rule the rest out.  Results meet in a
//            per-query 64-bit LDS atomic min on (distance bits, position).  Ring 1 (the 3 x 3 window rows) first; a query that
//            is not provably exact after it queues the listed rows that can still hold a closer point; rings beyond the list
//            use the per-query shell walk;
//   tail     lanes 0..31, one query each, FP64: K4 error under the previous correspondences, gate, Mahalanobis,
//            residual / Jacobian / normal equations.  Its operands were requested before the search.
// Target points are NOT staged in LDS: measured on MI355X, copying a region's rows (LDS-DMA, 640 points per wave) cost
// more than it saved - the 8 MB target lives in L2 / Infinity Cache, a walk touches 8-16 points of a row, and the LDS
// the copies needed held occupancy at 8 waves per CU.  What pays is the row LIST (which rows exist, where they start)
// and 12 waves per CU hiding the walks' latency.
// Partial sums are stored per GROUP ([group][slot]: one 256-byte row per block), so that the result does not depend on the
// order in which the groups are launched - which the solver sorts by measured cost.
constexpr int kStageRowsPerLane = 4;
constexpr int kStageRows = 64 * kStageRowsPerLane;  // (y,z) rows of a batch region
constexpr int kStageXs = 20;         // cells per row of the region
constexpr int kStageMaxGrow = 6;

#ifndef NGICP_PASS_WAVES
#define NGICP_PASS_WAVES 3  // waves per SIMD the pass kernel is compiled for by default.  4 fits (128 VGPRs with 6 spilled dwords, 40 KB LDS per
                            // block) and was measured: c3 +5 %, c2 +6 %, c5 -5 % in time - twelve waves already saturate a CU's gather path on
                            // a grid of one round of blocks; the host launches the 4-wave build on grids of more than two rounds
#endif
#ifndef NGICP_WALK_WINDOW
#define NGICP_WALK_WINDOW 12
#endif

#ifndef NGICP_WALK_PREFETCH
#define NGICP_WALK_PREFETCH 0  // (measured: c3 32.8 -> 35.8 us, c5 39.2 -> 41.2: the extra requests cost more than the warm lines save) 0: none; 1: the adjacent point of the next window(s); 2: their far end (scan_global_outward)
#endif
#ifndef NGICP_PRE_FAR
#define NGICP_PRE_FAR 0  // (measured: exact, no faster - c3 34.1 vs 33.7 us, c5 43.1 vs 41.3) listed rows queued together with the ring-1 units when the warm start says they will be needed (ngicp_pass_group.inc)
#endif
#ifndef NGICP_ALIGNED_WINDOWS
#define NGICP_ALIGNED_WINDOWS 0
#endif
constexpr int kWalkWindow = NGICP_WALK_WINDOW;  // points per walk window (one memory round trip); <= kSortedPad
static_assert(kWalkWindow <= kSortedPad && kWalkWindow % 2 == 0, "walk windows may overhang the array by at most the sentinel frame");

constexpr int kUnitCap = 288;         // queued (query, row) walks per round (ring 1 needs 32 queries x 9 rows)

struct WaveStage {
  union {
    struct {
      int4 live[kStageRows];  // one record per NON-EMPTY row of the region, nearest ring first: {y | z << 16, points, first point, -}
      int qstat[32][3];       // diagnostic counters (PassArgs::dbg_qstats)
    };
    double red[16 * 30];  // the per-batch reduction reuses the (then idle) tables as scratch, sixteen queries at a time
  };
  // Work distribution inside the wave: a (query, row) walk is a UNIT.  Units are queued here and popped by whichever lane is
  // free, so a query that needs twenty rows is served by twenty lanes instead of its own two while the lanes of queries that
  // needed one row would idle.  Results meet in qkey by a 64-bit atomic min on (distance bits, position): the total order.
  unsigned long long qkey[32];  // per query: nearest so far
  float4 qtab[32];              // per query: transformed coordinates
  // a unit: ring 1: {query | row code << 5 | (start - run start) << 9, run start, run end, (y,z) gap}; beyond: {query | listed row << 5}
  int unit_q[kUnitCap], unit_s[kUnitCap], unit_e[kUnitCap];
  float unit_g[kUnitCap];
  int q_tail, q_head;
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Outward scan of an x-sorted run [s, e) in GLOBAL memory from a starting guess m: walk right, then left, each until the
// x-gap alone rules the rest out.  Any start is correct (a side only stops once it is past qx AND out of reach); a good
// start (interpolated from the cell geometry: points of a dense scan line are nearly equidistant in x) makes the cost
// O(points within reach) with no search at all.  This is what keeps dense scan lines affordable.
template <int W = kWalkWindow, int kStep = kWalkWindow, class PT>
__device__ __forceinline__ void scan_global_outward(const PT* __restrict__ tgt, int s, int e, int m, float qx, float qy, float qz, float gyz, float gate_sq,
                                                    float& best, int& pos, unsigned int& ncand, unsigned int& gsteps) {
  if (e <= s) return;
  gsteps += 0x10000u;
  m = min(max(m, s), e - 1);
  // ONE loop body serves the three kinds of step (the 8 points around the start, then 8 more to the right while that side is
  // alive, then to the left): a window [w, w + 8) of the run, always read in increasing position, so that c[0] / c[7] are its
  // smallest / largest x and, among equal distances, the first one met has the smallest position (strict `<` below).
  // Most walks end after the first window: both neighbours are ruled out by their x-gap alone.
  // Packed 12-byte points are read FOUR AT A TIME as three 16-byte loads (windows begin at a multiple of four points, i.e. of 48
  // bytes; the array and its sentinel frame are 16-byte aligned): 12 load instructions per 16-point window instead of 16.  What a
  // launch queues on is the CU's address path, per wave instruction (scripts/micro/vmem_issue.hip).
  constexpr bool kQuadLoads = NGICP_ALIGNED_WINDOWS && sizeof(PT) == 12 && W % 4 == 0 && kStep % 4 == 0;
  const int w0 = kQuadLoads ? ((m - W / 2) & ~3) : m - W / 2;
  int w = w0, dir = 0, lo = w0, hi = w0 + W;  // [lo, hi) has been read
  int wn = W;  // points of the current window: W around the start, kStep on either side after that.  Measured with W = 12: side windows
               // of 8 / 6 / 4 points examine 8 / 12 / 16 % fewer candidates and cost c3 +3 / +7 / +18 %, c2 +1 / +6 / +16 % in time (more
               // dependent steps) but c5 -3 % (8 points): the build for large grids, bound by what its waves fetch, uses 8
  static_assert(kStep >= 2 && kStep <= W, "side windows are at most as long as the first");
  bool go_left = false;
  for (;;) {
    // No index clamps: a window that overhangs [s, e) reads points of the neighbouring runs (genuine target points: they can
    // only be legitimate candidates) or the sentinels that frame the array (infinitely far).  The x-gap tests below only
    // look at the window's last / first point when its right / left end is inside the run.
    const PT* __restrict__ q = tgt + w;  // one address, immediate offsets
    PT c[W];
    if constexpr (kQuadLoads) {
      const float4* __restrict__ q4 = reinterpret_cast<const float4*>(q);
      float4 r[3 * W / 4];
#pragma unroll
      for (int j = 0; j < 3 * W / 4; ++j)
        if (j < 3 * kStep / 4 || wn == W) r[j] = q4[j];
#pragma unroll
      for (int j = 0; j < W / 4; ++j) {
        if (j < kStep / 4 || wn == W) {
          const float4 r0 = r[3 * j], r1 = r[3 * j + 1], r2 = r[3 * j + 2];
          c[4 * j].x = r0.x; c[4 * j].y = r0.y; c[4 * j].z = r0.z;
          c[4 * j + 1].x = r0.w; c[4 * j + 1].y = r1.x; c[4 * j + 1].z = r1.y;
          c[4 * j + 2].x = r1.z; c[4 * j + 2].y = r1.w; c[4 * j + 2].z = r2.x;
          c[4 * j + 3].x = r2.y; c[4 * j + 3].y = r2.z; c[4 * j + 3].z = r2.w;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (j < kStep || wn == W) c[j] = q[j];
    }
    // Software prefetch: one word of the window(s) the walk would read NEXT (after the first window: the one on either side; later: the
    // next in the walk's direction), requested together with this window and looked at only after it has been worked on.  Four in ten
    // walks go on to a second window, and a step whose line is already in the L2 is a third of a step that goes out to memory.
    unsigned int pf = 0;
    if constexpr (NGICP_WALK_PREFETCH != 0) {
      if (dir >= 0 && hi < e) pf ^= *reinterpret_cast<const unsigned int*>(tgt + (hi + (NGICP_WALK_PREFETCH > 1 ? kStep - 1 : 0)));
      if (dir <= 0 && lo > s) pf ^= *reinterpret_cast<const unsigned int*>(tgt + (lo - (NGICP_WALK_PREFETCH > 1 ? kStep : 1)));
    }
    float lb = sqdist(qx, qy, qz, c[0]);
    int lj = 0;
#pragma unroll
    for (int j = 1; j < W; ++j) {
      if (j < kStep || wn == W) {
        const float d = sqdist(qx, qy, qz, c[j]);
        if (d < lb) { lb = d; lj = j; }
      }
    }
    if (nn_better(lb, w + lj, best, pos)) { best = lb; pos = w + lj; }
    ncand += wn;
    ++gsteps;
    if constexpr (NGICP_WALK_PREFETCH != 0) gsteps += (pf == 0x7fc0beefu) ? 0x100u : 0u;  // (keeps the prefetched word alive until here; gsteps is a diagnostic counter)
    const float lim = fminf(best, gate_sq), dr = (wn == W ? c[W - 1].x : c[kStep - 1].x) - qx, dl = qx - c[0].x;
    const bool more_right = hi < e && !(dr > 0.f && dr * dr + gyz > lim);
    const bool more_left = lo > s && !(dl > 0.f && dl * dl + gyz > lim);
    if (dir == 0) go_left = more_left;
    if (dir >= 0 && more_right) {
      dir = 1;
      w = hi;
      hi += kStep;
      wn = kStep;
    } else if (dir >= 0 ? go_left : more_left) {
      dir = -1;
      lo -= kStep;
      w = lo;
      wn = kStep;
    } else {
      break;
    }
  }
}

// The same walk done by the FOUR lanes of a quad on one run: lane ql of the quad reads the points w + ql, w + ql + 4, ... of a window,
// so one load instruction of the quad covers four consecutive points (48 contiguous bytes) and a W-point window is W / 4 wave
// instructions instead of W.  The CU's address path is what a launch queues on (a scattered gather costs it ~20 cycles per wave
// instruction plus ~0.7 per distinct line, scripts/micro/vmem_issue.hip), and a dependent step of a walk is as long as the
// instructions of the waves queued ahead of it: four times fewer of them per window.  All control flow is quad-uniform; (best, pos)
// come back identical on the four lanes.  The winner is the same as scan_global_outward's: smallest (distance, position).
#ifndef NGICP_WALK_LANES
#define NGICP_WALK_LANES 1
#endif
constexpr int kWalkLanes = NGICP_WALK_LANES;  // 1: a lane walks a run by itself; 4: a quad does
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float quad_perm_f(float v) { return __int_as_float(quad_perm_i<CTRL>(__float_as_int(v))); }
template <int W, int kStep, class PT>
__device__ __forceinline__ void scan_quad_outward(const PT* __restrict__ tgt, int s, int e, int m, float qx, float qy, float qz, float gyz, float gate_sq,
                                                  float& best, int& pos, unsigned int& ncand, unsigned int& gsteps, int ql) {
  static_assert(W % 4 == 0 && kStep % 4 == 0 && kStep >= 4 && kStep <= W, "windows are dealt to the four lanes of a quad");
  constexpr int PER = W / 4, PERS = kStep / 4;
  if (e <= s) return;
  gsteps += 0x10000u;
  m = min(max(m, s), e - 1);
  int w = m - W / 2, dir = 0, lo = m - W / 2, hi = m - W / 2 + W;  // [lo, hi) has been read
  int wn = W;
  bool go_left = false;
  for (;;) {
    const PT* __restrict__ q = tgt + (w + ql);  // (no index clamps: see scan_global_outward)
    PT c[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (j < PERS || wn == W) c[j] = q[4 * j];
    float lb = sqdist(qx, qy, qz, c[0]);
    int lj = 0;
#pragma unroll
    for (int j = 1; j < PER; ++j) {
      if (j < PERS || wn == W) {
        const float d = sqdist(qx, qy, qz, c[j]);
        if (d < lb) { lb = d; lj = j; }
      }
    }
    int lp = w + ql + 4 * lj;
    // the window's first / last x: lane 0's first point, lane 3's last
    const float xl = quad_perm_f<0x00>(c[0].x), xr = quad_perm_f<0xff>(wn == W ? c[PER - 1].x : c[PERS - 1].x);
    {
      const float od = quad_perm_f<0xb1>(lb);
      const int op = quad_perm_i<0xb1>(lp);
      if (nn_better(od, op, lb, lp)) { lb = od; lp = op; }
    }
    {
      const float od = quad_perm_f<0x4e>(lb);
      const int op = quad_perm_i<0x4e>(lp);
      if (nn_better(od, op, lb, lp)) { lb = od; lp = op; }
    }
    if (nn_better(lb, lp, best, pos)) { best = lb; pos = lp; }
    if (ql == 0) ncand += wn;
    ++gsteps;
    const float lim = fminf(best, gate_sq), dr = xr - qx, dl = qx - xl;
    const bool more_right = hi < e && !(dr > 0.f && dr * dr + gyz > lim);
    const bool more_left = lo > s && !(dl > 0.f && dl * dl + gyz > lim);
    if (dir == 0) go_left = more_left;
    if (dir >= 0 && more_right) {
      dir = 1;
      w = hi;
      hi += kStep;
      wn = kStep;
    } else if (dir >= 0 ? go_left : more_left) {
      dir = -1;
      lo -= kStep;
      w = lo;
      wn = kStep;
    } else {
      break;
    }
  }
}

// Outer shells (Chebyshev radius > rdone) without a row list; continues from the (best, pos) found in rings 0..rdone
// until the exactness bound or the distance gate ends the search.  Rows whose (y,z) gap already exceeds the reach are skipped
// without a memory access; the bounds of the others are fetched in chunks of 4 rows before any row is walked (most shell
// rows are empty, so a lane pays one memory round trip per chunk instead of one per row); the walks are x-pruned like
// everywhere else, so a large distance gate costs O(R^2) rows per shell, not O(R^3) points.
template <int G, class PT>
__device__ __forceinline__ void nn_shells(const Grid& g, const PT* __restrict__ tgt, const int* __restrict__ cell_start, float qx, float qy, float qz,
                                          int cx, int cy, int cz, float gate_sq_f, int sub, int rdone, float& best, int& pos, unsigned int& ncand) {
  const int rmax = max(max(g.nx, g.ny), g.nz);
  unsigned int steps = 0;
  for (int r = rdone;; ++r) {
    const float bound = unexplored_bound_sq(g, qx, qy, qz, cx, cy, cz, r);
    if (best <= bound || bound >= gate_sq_f || r >= rmax) break;
    // shell r + 1: rows t = sub, sub + G, ... of the (2R+1)^2 (y,z) window clipped to the grid
    const int R = r + 1;
    const int z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
    const int xa = max(cx - R, 0), xb = min(cx + R, g.nx - 1);
    const bool lo = cx - R >= 0, hi = cx + R <= g.nx - 1;
    const int wy = y1 - y0 + 1;
    const int nrows = wy * (z1 - z0 + 1);
    const float xfrac = fminf(fmaxf((qx - (g.ox + (float)xa * g.h)) / ((float)(xb + 1 - xa) * g.h), 0.f), 1.f);
    for (int t0 = sub; t0 < nrows; t0 += 4 * G) {
      int s0[4], e0[4], s1[4], e1[4];
      float gap[4];
      const float lim = fminf(best, gate_sq_f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * G;
        const bool ok = t < nrows;
        const int z = z0 + (ok ? t : 0) / wy, y = y0 + (ok ? t : 0) % wy;
        const int row = (z * g.ny + y) * g.nx;
        const bool face = z == cz - R || z == cz + R || y == cy - R || y == cy + R;
        gap[u] = row_gap_sq(g, y, z, cy, cz, qy, qz);
        const bool want = ok && gap[u] <= lim;
        // face rows: one run [xa, xb] (marked by s1 = -1); interior rows: the two end cells
        const int a0 = face ? xa : cx - R, a1 = face ? xb + 1 : cx - R + 1;
        const bool use0 = want && (face || lo), use1 = want && !face && hi;
        s0[u] = use0 ? cell_start[row + a0] : 0;
        e0[u] = use0 ? cell_start[row + a1] : 0;
        s1[u] = use1 ? cell_start[row + cx + R] : (face ? -1 : 0);
        e1[u] = use1 ? cell_start[row + cx + R + 1] : 0;
      }
#pragma unroll 1
      for (int u = 0; u < 4; ++u) {
        int a0 = s0[0], b0 = e0[0], a1 = s1[0], b1 = e1[0];
        float gp = gap[0];
#pragma unroll
        for (int v = 1; v < 4; ++v)
          if (u == v) a0 = s0[v], b0 = e0[v], a1 = s1[v], b1 = e1[v], gp = gap[v];
        if (gp > fminf(best, gate_sq_f)) continue;
        // a face row starts where qx sits in it; the end cells start at their end nearest to the query
        if (b0 > a0) scan_global_outward(tgt, a0, b0, a1 < 0 ? a0 + (int)(xfrac * (float)(b0 - a0)) : b0 - 1, qx, qy, qz, gp, gate_sq_f, best, pos, ncand, steps);
        if (b1 > a1 && a1 >= 0) scan_global_outward(tgt, a1, b1, a1, qx, qy, qz, gp, gate_sq_f, best, pos, ncand, steps);
      }
    }
    if (G > 1) group_min<G>(best, pos);
  }
}


// Upper-triangular packing used by the pass: index of (r,c), r <= c, in the 21-vector
__host__ __device__ __forceinline__ int tri21(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

// --- the solver ----------------------------------------------------------------------------------
constexpr int kMaxOrderGroups = 4096;  // launch-order sort: groups whose costs fit the solver's LDS and one load round (8 per thread)
constexpr int kSolveThreads = 512;  // 8 waves = 2 per SIMD: the serial lane keeps a 256-VGPR budget, the reduction gets 512 loaders
constexpr int kSolveSubs = kSolveThreads / 16;  // sub-sums per slot pair (thread = 16 slot pairs x 32 group subsets)
#ifndef NGICP_SOLVE_CHUNK
#define NGICP_SOLVE_CHUNK 40
#endif
constexpr int kSolveChunk = NGICP_SOLVE_CHUNK;                  // 16-byte loads a thread keeps in flight per step: one step covers 32 x 40 = 1280 groups

constexpr int kSolveRowSubsets = 32;  // the solver adds the group rows in 32 subsets (row g belongs to subset g mod 32), then the subsets in order
constexpr int kProgressDone = 1 << 30, kProgressMask = kProgressDone - 1;

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
  L.staged_total += sums[31];

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

// One step of the optimiser's state machine with the sums of a finished pass (the serial lane of the solver; k_gicp_head's blocks run it
// redundantly, each for itself): lm_advance, and on an accepted trial the bookkeeping of LsqRegistration::align's outer loop.
__device__ __forceinline__ void lm_step(LmHot& L, const LmConfig& cfg, const double* sums, double* trace, int max_trace_rows) {
  const bool gn = cfg.optimizer == 0;
  const bool accepted = lm_advance(L, cfg, sums, trace, max_trace_rows);
  if (accepted) {
#pragma unroll
    for (int i = 0; i < 36; ++i) L.final_H[i] = L.H[i];
    if (!gn) {
      // LM accept: x0 = xi, lambda update, convergence, next outer iteration (impl/lsq_registration_impl.hpp:201-204,110)
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
}

// The solver's block-shared scratch.  k_lm_solve owns one; the fused tail of k_gicp_pass lays it over the (then idle) search tables.
template <int THREADS>
struct SolveShared {
  double wsum[32][kPartialStride];
  double sums[kPartialStride];
  int ord_cnt[THREADS / 64 - 1][16], ord_pos[THREADS / 64 - 1][16];  // per sorting wave and cost class
  int ord_cost[kMaxOrderGroups];
  unsigned char ord_cls[kMaxOrderGroups];
  int ord_arrived;
  LmHot L;
};

// The body of the solver for a block of THREADS threads (512: k_lm_solve; 256: the last block of a fused pass).  AGENT: the group rows
// and costs were written by other blocks of the SAME launch (write-through stores, published by a ticket; the caller has run the
// agent-scope acquire): they are then read with agent-scope loads, which bypass this CU's L1 whatever it holds.
// PERSIST: the solver between two passes of the persistent kernel.  There is no kernel boundary and no fence anywhere: every word another
// block wrote (group rows, costs - and the state image, which the solver of the previous pass, possibly on another XCD, stored) is read
// with agent-scope loads, the state image and the pass's view of it are stored write-through.
template <int THREADS, bool AGENT, bool PERSIST = false>
__device__ __forceinline__ void lm_solve_body(const SolveArgs& a, SolveShared<THREADS>& sh) {
  static_assert(!PERSIST || AGENT, "the persistent solver is an agent-scope one");
  constexpr int kSolveThreads = THREADS;
  constexpr int kSolveSubs = 32;                      // subsets of rows (row g belongs to subset g mod 32), whatever the block size
  constexpr int kThreadSubs = THREADS / 16;           // subsets the block's threads cover at once
  constexpr int kPer = kSolveSubs / kThreadSubs;      // subsets per thread (512 threads: 1, 256 threads: 2)
  constexpr int kSolveChunk = THREADS >= 512 ? NGICP_SOLVE_CHUNK : 14;  // 16-byte loads per subset a thread keeps in flight per step (4 VGPRs each; the fused block has the pass kernel's 168-register budget)
  static_assert(kPer >= 1 && kPer * kThreadSubs == kSolveSubs, "block sizes: 256 or 512 threads");
  auto& wsum = sh.wsum;
  auto& sums = sh.sums;
  auto& ord_cnt = sh.ord_cnt;
  auto& ord_pos = sh.ord_pos;
  auto& ord_cost = sh.ord_cost;
  auto& ord_cls = sh.ord_cls;
  int& ord_arrived = sh.ord_arrived;
  LmHot& L = sh.L;
  if (threadIdx.x == 0) ord_arrived = 0;  // (visible to every wave after the barriers below)
  LmState* st = a.st;
  LmState* sto = a.st_out ? a.st_out : a.st;
#define NG_SSTAMP(k)                                                                 \
  do {                                                                               \
    if (a.dbg_stamps && threadIdx.x == 0) a.dbg_stamps[k] = __builtin_amdgcn_s_memtime(); \
  } while (0)
  // (as a VECTOR load - an atomic load is never made a scalar one: scalar loads return out of order, so the wait for the kernel arguments the
  // row addresses need would also have waited for this cold word before the first row load could go out.  Persistent: the blocks left
  // the pass loop when they saw the flag.)
  const int done_at_entry = PERSIST ? 0 : __hip_atomic_load(&st->hot.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const unsigned long long t_solver_entry = a.dbg_stamps ? __builtin_amdgcn_s_memtime() : 0ull;  // (stored below, once the flag has been looked at:
                                                                                                 // looking at it HERE put a cold scalar round trip in front of every load of the block)
  // ---- deterministic reduction of the group partials.  Thread = (slot pair vp, group subset sb): it adds the rows
  //      sb, sb + 32, sb + 64, ... of its two slots in increasing order, kSolveChunk sixteen-byte loads in flight per step (one step
  //      covers 1280 groups: the whole c3 grid in a single memory round trip, c5 in two); the 32 subset sums of a slot are then added
  //      in subset order.  Fixed order throughout: bit-reproducible, independent of the launch order of the pass. ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vp = threadIdx.x & 15, sb = threadIdx.x >> 4;
  const double2* __restrict__ rows = reinterpret_cast<const double2*>(a.partials) + vp;  // row g: rows[g * 16]
  // Everything the block needs from memory is requested HERE, before the first wait: the state image the serial lane will work on
  // (fetched by the whole block, one coalesced access, instead of by lane 0 after the reduction), the first step of group rows, the
  // groups' measured costs.  Consumed one after the other they were three dependent round trips of ~0.9 us each.
  static_assert(sizeof(LmHot) % 4 == 0, "LmHot is copied as dwords");
  constexpr int kHotWords = (int)(sizeof(LmHot) / 4), kHotPerThread = (kHotWords + kSolveThreads - 1) / kSolveThreads;
  constexpr int kCostPerThread = kMaxOrderGroups / kSolveThreads;
  // (the fused block has three sorting waves instead of seven: the order is refreshed after the first three passes of an alignment -
  // the costs settle with the warm start - and after every fourth from then on, so that it stays off the serial lane's path)
  // (persistent: the blocks keep their groups for the whole alignment; the order built from the costs of the third pass - warm starts in
  // place - goes to the buffer the NEXT alignment launches with)
  const int passes_at_entry = PERSIST ? __hip_atomic_load(&st->hot.passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : st->hot.passes;
  const bool order_it = a.mode == 0 && a.grp_order && a.nblocks <= kMaxOrderGroups && (PERSIST ? passes_at_entry == 2 : (!AGENT || passes_at_entry < 3 || (passes_at_entry & 3) == 3));
  int hv[kHotPerThread];
#pragma unroll
  for (int k = 0; k < kHotPerThread; ++k) {
    const int w = threadIdx.x + k * kSolveThreads;
    if constexpr (PERSIST) hv[k] = w < kHotWords ? __hip_atomic_load(reinterpret_cast<const int*>(&st->hot) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    else hv[k] = w < kHotWords ? reinterpret_cast<const int*>(&st->hot)[w] : 0;
  }
  // (One CU moves 64 B per clock: the 222 KB of 866 rows are ~3.5k cycles on top of the latency.  Rows beyond the grid are skipped by
  // a branch each: fetching a stand-in row instead - branch-free issue - measured slower, the stand-ins pile up on one channel.)
  const int last_row = (a.nrows ? a.nrows : a.nblocks) - 1, last_grp = a.nblocks - 1;
  // (AGENT: the caller's agent-scope acquire has dropped this CU's L1, the producers stored write-through and no line of these rows
  // can be in this XCD's L2 from before - nobody read them earlier in this launch: plain 16-byte loads, as MI355X_MICROARCH.md's
  // "valid forms" prescribe for the consumer side)
  // (PERSIST: this XCD's L2 may hold the rows of an earlier pass - one of its blocks was the solver then - so the loads say agent scope)
  auto load_row = [](const double2* q) -> double2 {
    if constexpr (PERSIST) {
      const double* d = reinterpret_cast<const double*>(q);
      return make_double2(__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    } else {
      return *q;
    }
  };
  // (The order of the sums never depends on the block size: 32 subsets of rows; a thread of a 256-thread block takes two of them.)
  double2 p[kPer * kSolveChunk];
#pragma unroll
  for (int sI = 0; sI < kPer; ++sI)
#pragma unroll
    for (int j = 0; j < kSolveChunk; ++j) {
      const int gi = sb + sI * kThreadSubs + j * kSolveSubs;
      p[sI * kSolveChunk + j] = gi <= last_row ? load_row(rows + (size_t)gi * (kNumSlots / 2)) : make_double2(0.0, 0.0);
    }
  int oc[kCostPerThread];
#pragma unroll
  for (int k = 0; k < kCostPerThread; ++k) {
    const int gi = threadIdx.x + k * kSolveThreads;
    if constexpr (PERSIST) oc[k] = (order_it && gi <= last_grp) ? __hip_atomic_load(a.grp_cost + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    else oc[k] = (order_it && gi <= last_grp) ? a.grp_cost[gi] : 0;
  }
  if (a.mode == 0 && done_at_entry) return;  // (a scalar load issued at the top: it does not wait for the vector loads above)
  if (a.dbg_stamps && threadIdx.x == 0) a.dbg_stamps[0] = t_solver_entry;  // (working launches only)
  // The serial lane works on the LDS image of the state in place (a register-resident copy needs ~260 VGPRs: it spills at two
  // waves per SIMD), and wave 0 stores it back with one coalesced pass.
#pragma unroll
  for (int k = 0; k < kHotPerThread; ++k) {
    const int w = threadIdx.x + k * kSolveThreads;
    if (w < kHotWords) reinterpret_cast<int*>(&L)[w] = hv[k];
  }
  if (order_it) {  // the groups' costs, for the launch order built further down (visible after the barriers below)
#pragma unroll
    for (int k = 0; k < kCostPerThread; ++k) {
      const int gi = threadIdx.x + k * kSolveThreads;
      if (gi < a.nblocks) ord_cost[gi] = oc[k];
    }
  }
  {
    double a0[kPer], a1[kPer];
#pragma unroll
    for (int sI = 0; sI < kPer; ++sI) {
      a0[sI] = p[sI * kSolveChunk].x;  // (not 0.0 + p[0]: the compiler places that add, and a wait for the first load alone, before the other loads)
      a1[sI] = p[sI * kSolveChunk].y;
#pragma unroll
      for (int j = 1; j < kSolveChunk; ++j) {
        a0[sI] += p[sI * kSolveChunk + j].x;
        a1[sI] += p[sI * kSolveChunk + j].y;
      }
    }
    for (int g0 = kSolveSubs * kSolveChunk; g0 + sb <= last_row; g0 += kSolveSubs * kSolveChunk) {  // larger grids: further steps
#pragma unroll
      for (int sI = 0; sI < kPer; ++sI)
#pragma unroll
        for (int j = 0; j < kSolveChunk; ++j) {
          const int gi = g0 + sb + sI * kThreadSubs + j * kSolveSubs;
          p[sI * kSolveChunk + j] = gi <= last_row ? load_row(rows + (size_t)gi * (kNumSlots / 2)) : make_double2(0.0, 0.0);
        }
#pragma unroll
      for (int sI = 0; sI < kPer; ++sI)
#pragma unroll
        for (int j = 0; j < kSolveChunk; ++j) {
          a0[sI] += p[sI * kSolveChunk + j].x;
          a1[sI] += p[sI * kSolveChunk + j].y;
        }
    }
    NG_SSTAMP(1);
#pragma unroll
    for (int sI = 0; sI < kPer; ++sI) {
      wsum[sb + sI * kThreadSubs][2 * vp] = a0[sI];
      wsum[sb + sI * kThreadSubs][2 * vp + 1] = a1[sI];
    }
  }
  __syncthreads();
  if (threadIdx.x < kPartialStride) {
    const int v = threadIdx.x;
    double t = 0.0;
    if (v < kNumSlots)
      for (int sb = 0; sb < kSolveSubs; ++sb) t += wsum[sb][v];
    sums[v] = t;
  }
  __syncthreads();
  NG_SSTAMP(2);
  if (a.sums_out && threadIdx.x < kPartialStride) a.sums_out[threadIdx.x] = sums[threadIdx.x];
  if (a.mode == 3) return;  // reduce only (point-sharded stepping: the caller all-reduces sums_out)
  if (order_it && wave >= 1) {
    const unsigned long long t_ord = a.dbg_stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    // ---- launch order of the next pass, built by waves 1..7 (costs already in LDS) while lane 0 of wave 0 runs the state machine:
    //      a counting sort into 16 cost classes relative to the slowest group, heaviest class first, ascending group index inside
    //      a class.  (The pass's results do not depend on the launch order.) ----
    // A lone wave is bound by the latency of its own LDS round trips (~600 cycles per 64 groups and loop: 7 us for 866 groups, 17 us
    // for 2217 - longer than the state machine), so every wave sorts a contiguous SLICE of the groups: it counts its slice per class,
    // the seven waves meet at a counter in LDS (all waves of a block are resident: the wait cannot deadlock, and it is bounded), and
    // each then knows where its slice starts inside every class.  No per-lane atomics and no per-class loops: four ballots (one per
    // bit of the class) give every lane the mask of the lanes that share its class; its rank is a popcount, and the lowest lane of
    // each class moves the class's counter in LDS (distinct addresses per class; same-wave LDS operations are ordered).
    constexpr int kSortWaves = kSolveThreads / 64 - 1;
    const int sw = wave - 1;
    auto wsync = [] {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto same_class = [&](int cls, bool valid) -> unsigned long long {
      unsigned long long m = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 4; ++bit) {
        const bool one = (cls >> bit) & 1;
        const unsigned long long bb = __ballot(valid && one);
        m &= one ? bb : ~bb;
      }
      return m;
    };
    if (lane < 16) ord_cnt[sw][lane] = 0;
    int mx = 1;  // (every wave looks at all the costs: no exchange needed for the maximum)
    for (int gi = lane; gi < a.nblocks; gi += 64) mx = max(mx, ord_cost[gi]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    const float to_class = 16.0f / ((float)mx + 1.0f);  // (a heuristic: float rounding at class boundaries is immaterial)
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int slice = ((a.nblocks + kSortWaves * 64 - 1) / (kSortWaves * 64)) * 64;  // groups per wave, a multiple of 64
    const int g_first = sw * slice, g_last = min(g_first + slice, a.nblocks);
    wsync();
    for (int g0 = g_first; g0 < g_last; g0 += 64) {
      const int gi = g0 + lane;
      const bool valid = gi < g_last;
      const int cls = valid ? 15 - min(15, (int)((float)ord_cost[gi] * to_class)) : 0;
      if (valid) ord_cls[gi] = (unsigned char)cls;
      const unsigned long long m = same_class(cls, valid);
      if (valid && (m & lt) == 0) ord_cnt[sw][cls] += __popcll(m);  // the class's lowest lane
      wsync();
    }
    // the seven waves meet
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(&ord_arrived, 1);
    bool met = false;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      if (*reinterpret_cast<volatile int*>(&ord_arrived) >= kSortWaves) {
        met = true;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (met) {  // (never false in practice; the order of the previous refresh then simply stays)
      if (lane < 16) {  // where this wave's slice starts inside class `lane`
        int run = 0;
        for (int c = 0; c < lane; ++c)
          for (int w = 0; w < kSortWaves; ++w) run += ord_cnt[w][c];
        for (int w = 0; w < sw; ++w) run += ord_cnt[w][lane];
        ord_pos[sw][lane] = run;
      }
      wsync();
      for (int g0 = g_first; g0 < g_last; g0 += 64) {
        const int gi = g0 + lane;
        const bool valid = gi < g_last;
        const int cls = valid ? ord_cls[gi] : 0;
        const unsigned long long m = same_class(cls, valid);
        const int base = ord_pos[sw][cls];
        if (valid) a.grp_order[base + __popcll(m & lt)] = gi;  // within a class: ascending group index
        wsync();
        if (valid && (m & lt) == 0) ord_pos[sw][cls] = base + __popcll(m);
        wsync();
      }
      if (sw == 0 && lane == 0 && a.order_valid) *a.order_valid = 1;
    }
    if (a.dbg_stamps && sw == 0 && lane == 0) a.dbg_stamps[7] = __builtin_amdgcn_s_memtime() - t_ord;
  }
  if (wave != 0) return;
  if (a.mode == 2) {  // compute_error hook
    if (lane == 0) st->hot.y0 = sums[28];
    return;
  }
  if (lane == 0) {
    if (a.mode == 1) {  // linearize hook
      adopt_new(L, sums);
    } else {
      const LmConfig& cfg = a.cfg;
      NG_SSTAMP(3);
      lm_step(L, cfg, sums, a.trace, a.max_trace_rows);
      NG_SSTAMP(4);
      NG_SSTAMP(5);
      if (!L.done) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float f = (float)(c < 3 ? L.xi.R[r * 3 + c] : L.xi.t[r]);
            // (persistent: solvers of different passes sit on different XCDs - a plain store would leave one dirty copy of the word per L2)
            if constexpr (PERSIST) __hip_atomic_store(&st->xi_f[r * 4 + c], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else sto->xi_f[r * 4 + c] = f;
          }
        }
      }
    }
  }
  // wave 0 stores the state image back (lane 0's LDS writes are ordered before the other lanes' reads by the fence pair)
  if (a.mode == 0 && lane == 0 && L.done) {
    L.t_done = __builtin_amdgcn_s_memrealtime();
    if (a.t_first) L.t_first = PERSIST ? __hip_atomic_load(a.t_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *a.t_first;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool finished_now = a.mode == 0 && L.done != 0;
  // The end of an alignment goes to the host WITHOUT a copy or a stream synchronisation: the image is written to pinned memory
  // here, then (system-scope release) the done flag; the host, which polls that word anyway, reads the image as soon as it sees
  // it - the launches it had enqueued ahead return at once behind its back.
  if (finished_now && a.final_host) {
    for (int w = lane; w < (int)(sizeof(LmHot) / 4); w += 64)
      __hip_atomic_store(reinterpret_cast<int*>(a.final_host) + w, reinterpret_cast<const int*>(&L)[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
  }
  // progress for the host (it keeps a few (pass, solve) pairs in flight and stops feeding the stream when it sees the flag)
  if (a.mode == 0 && a.progress_host && lane == 0)
    __hip_atomic_store(a.progress_host, (L.passes & kProgressMask) | (L.done ? kProgressDone : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (ordered behind the image by the fence above when done)
  if constexpr (PERSIST) {
    for (int w = lane; w < (int)(sizeof(LmHot) / 4); w += 64)
      __hip_atomic_store(reinterpret_cast<int*>(&st->hot) + w, reinterpret_cast<const int*>(&L)[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the next pass's view of the state (LmState::view)
    int v = 0;
    if (lane < 24) {
      v = reinterpret_cast<const int*>(&L.xi)[lane];
    } else if (lane < 36) {
      const int r = (lane - 24) >> 2, cI = (lane - 24) & 3;
      v = __float_as_int((float)(cI < 3 ? L.xi.R[r * 3 + cI] : L.xi.t[r]));
    } else if (lane == kViewDone) {
      v = L.done;
    } else if (lane == kViewCur) {
      v = L.cur;
    } else if (lane == kViewHaveLin) {
      v = L.have_lin;
    }
    __hip_atomic_store(st->view + (size_t)L.passes * kViewWords + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (pass L.passes comes next)
  } else {
    for (int w = lane; w < (int)(sizeof(LmHot) / 4); w += 64) reinterpret_cast<int*>(&sto->hot)[w] = reinterpret_cast<const int*>(&L)[w];
    // (a state that ends an alignment keeps the float pose of its last pass: k_corr_to_original reads it from wherever the state lies)
    if (sto != st && L.done && lane < 12) sto->xi_f[lane] = st->xi_f[lane];
  }
  NG_SSTAMP(6);
#undef NG_SSTAMP
}

__global__ void __launch_bounds__(kSolveThreads) k_lm_solve(SolveArgs a) {
  __shared__ SolveShared<kSolveThreads> sh;
  lm_solve_body<kSolveThreads, false>(a, sh);
}

constexpr int kStampStride = 24;
#define NG_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (a.dbg_stamps && lane == 0) a.dbg_stamps[(size_t)(blockIdx.x * 4 + wave) * kStampStride + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

template <int G, int WPS = NGICP_PASS_WAVES, bool FUSED = false>
__global__ void __launch_bounds__(256, WPS) k_gicp_pass(PassArgs a) {
  constexpr int B = 64 / G;  // queries per wave batch
  // Walk windows (see scan_global_outward), measured with the packed points: the default build, whose launches are as long as their
  // slowest chain of window steps, does best with 16-point windows (c3 35.4 -> 34.3 us, c2 32.0 -> 31.2 us against 12; 18 and more
  // cost registers and time); the build for large grids, bound by what its waves fetch, with 12 points and 8-point side windows.
  constexpr int kWin = WPS >= 4 ? 12 : 16, kSideStep = WPS >= 4 ? 8 : 16;
  static_assert(kWin <= kSortedPad, "walk windows may overhang the array by at most the sentinel frame");
  static_assert(B == kBatchQueries, "query batches are built for 32 queries (2 lanes per query)");
  __shared__ double lds[4][kNumSlots];
  // (FUSED is a build of its own: the solver's reduction keeps 28 sixteen-byte loads in flight per thread, and with it in the kernel
  // the 4-waves-per-SIMD build spilled 35-38 registers instead of 2 - c5 41.6 -> 48.3 us per pass with the tail never even executed)
  struct Nothing {};
  union PassShared {
    WaveStage stage[4];
    typename std::conditional<FUSED, SolveShared<256>, Nothing>::type sv;  // the fused solver (the last block of the grid) works where the search tables were
  };
  __shared__ PassShared shm;
  __shared__ int last_block;
  WaveStage* stage_all = shm.stage;
  const LmState* __restrict__ st = a.st;
  // Everything a block needs before it can fetch a point - the done flag, the launch-order flag, its position's group - is requested in
  // ONE go, as scalar loads through the constant address space (none of it changes while the launch runs): the compiler had made six
  // dependent round trips of it (done -> have_lin -> cur -> pose -> order flag -> order entry, the last two as vector loads), ~4 us on
  // every block's path before its first point.
  typedef const int __attribute__((address_space(4))) * ConstIntPtr;
  const int done_now = st->hot.done, have_lin_now = st->hot.have_lin;
  const int cur = st->hot.cur, nxt = cur ^ 1;
  const int order_is_valid = *(ConstIntPtr)(unsigned long long)a.order_valid;             // (both pointers are set for every launch: prepare_loop)
  const int order_entry = ((ConstIntPtr)(unsigned long long)a.grp_order)[blockIdx.x];    // (read whether valid or not: the buffer exists)
  // trial pose (FP64) and its float cast
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = st->hot.xi.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = st->hot.xi.t[i];
  float Tf[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) Tf[i] = st->xi_f[i];
  // (all of the above is in flight before the first of it is looked at)
  asm volatile("" ::"s"(done_now), "s"(have_lin_now), "s"(cur), "s"(order_is_valid), "s"(order_entry), "s"(Tf[0]), "s"(Tf[11]), "s"(R[0]), "s"(t[2]));
  if (!(a.mode & 4) && done_now) return;

  const bool do_err = (a.mode & 1) && have_lin_now;
  const bool do_lin = (a.mode & 2);
  const float4* __restrict__ tpt_old = a.tpt[cur];
  const double* __restrict__ mahal_old = a.mahal[cur];
  float4* __restrict__ tpt_new = a.tpt[nxt];
  double* __restrict__ mahal_new = a.mahal[nxt];
  const Grid& g = a.grid;

  double wave_total = 0.0;  // lane v (< 29) accumulates slot v of this wave over its batches
  unsigned int ncand = 0, nvalid = 0, nstaged = 0;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (the wave's number in a scalar register: its batch record is then a scalar load)
  const int sub = lane % G, grp = lane / G;
  WaveStage& S = stage_all[wave];
  NG_STAMP(0);
  // A block owns one GROUP of four consecutive batches (one per wave); its partial sums are stored under the group's
  // index, so the result does not depend on the order in which groups are launched.  That order is the solver's business:
  // it sorts the groups by the duration measured in the previous pass, heaviest first (the grid is ~1.7 waves of blocks
  // deep, and the slowest groups take 2-3x the median: started late they would set the kernel's length).
  const int group = order_is_valid ? order_entry : (int)blockIdx.x;
  const unsigned long long t_start = a.grp_cost ? __builtin_amdgcn_s_memtime() : 0ull;
  if (a.dbg_span && threadIdx.x == 0) a.dbg_span[(size_t)blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime();  // (the 100 MHz counter: the same on every CU)
  if (a.t_first && blockIdx.x == 0 && threadIdx.x == 0 && !have_lin_now) *a.t_first = __builtin_amdgcn_s_memrealtime();
#define NG_HAVE_LIN have_lin_now
#include "ngicp_pass_group.inc"
#undef NG_HAVE_LIN
  if constexpr (FUSED) {
  // ---- the solver in the tail of the launch (R0's final sum, impl/nano_gicp_impl.hpp:260-267, and LsqRegistration's step,
  //      impl/lsq_registration_impl.hpp:161-208).  Every store of this block that another block will read was made by wave 0 as a
  //      write-through store; wave 0 drains them, then ONE lane takes a ticket (relaxed, agent scope).  The block that draws the
  //      last ticket of the grid knows that every other block's rows are in memory: one agent-scope acquire (drops this CU's L1;
  //      no other block pays anything), then it reads the rows with agent-scope loads, reduces them in the fixed group order and
  //      advances the optimiser exactly as k_lm_solve does.  (Round 2's version released with a fence per block: an agent-scope
  //      release writes the XCD's whole dirty L2 back - the pass's own 6 MB of outputs, once per block: 36 -> 143 us.) ----
  if (wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const int tk = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_block = (tk == (int)gridDim.x - 1) ? 1 : 0;
    }
  }
  __syncthreads();
  if (!last_block) return;
  if (threadIdx.x == 0) {
    __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch: ordered by the kernel boundary)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  lm_solve_body<256, true>(a.sa, shm.sv);
  }
}

// One group of one pass for the PERSISTENT kernel, which calls it once per group and pass (a function of its own: inlined into that
// kernel's loops, everything the loops keep alive pushed the search's registers into scratch memory - 2.5x the time per pass).  The
// arguments are read where the hardware put them, through the constant address space (a.field is then a scalar load the compiler may
// repeat instead of keeping the value, exactly as in a kernel of its own); the trial pose comes from entry pass_no of the ring of views.
// RING false (k_gicp_queue, one launch per pass): the pose comes from the state itself, which does not change while the launch runs, and
// the block's row is stored plainly - the solver launch reads it behind a kernel boundary.
template <int G, int WPS, bool RING, class A>
__device__ __forceinline__ void persist_group_body(A& a, const int group, const int pass_no, WaveStage* stage_all, double (*lds)[kNumSlots]) {
  constexpr bool FUSED = RING;  // (persistent: the block's row and cost are stored write-through)
  constexpr int B = 64 / G;
  constexpr int kWin = WPS >= 4 ? 12 : 16, kSideStep = WPS >= 4 ? 8 : 16;  // (see k_gicp_pass)
  static_assert(kWin <= kSortedPad && B == kBatchQueries, "see k_gicp_pass");
  const bool do_lin = (a.mode & 2);
  Grid g;  // (field by field: a reference to a generic Grid cannot bind to the constant address space)
  g.ox = a.grid.ox; g.oy = a.grid.oy; g.oz = a.grid.oz;
  g.h = a.grid.h; g.inv_h = a.grid.inv_h;
  g.nx = a.grid.nx; g.ny = a.grid.ny; g.nz = a.grid.nz;
  g.ncells = a.grid.ncells;
  g.slack = a.grid.slack;
  // the pass's view of the state: entry pass_no of the ring (see LmState::view) - it does not change while the pass runs
  typedef const int __attribute__((address_space(4))) * ViewPtr;
  typedef const double __attribute__((address_space(4))) * ViewPtrD;
  typedef const float __attribute__((address_space(4))) * ViewPtrF;
  typedef const LmState __attribute__((address_space(4))) * StatePtr;
  StatePtr st4 = (StatePtr)(unsigned long long)a.st;
  ViewPtr vw = (ViewPtr)(unsigned long long)(a.st->view + (size_t)(a.first_pass + pass_no) * kViewWords);
  const int have_lin = RING ? vw[kViewHaveLin] : st4->hot.have_lin;
  const int cur = (RING ? vw[kViewCur] : st4->hot.cur) & 1, nxt = cur ^ 1;  // (indices into two-element arrays of pointers, whatever the memory holds)
  ViewPtrD vx = RING ? (ViewPtrD)(vw + kViewXi) : (ViewPtrD)&st4->hot.xi;  // (Pose: R[9] then t[3], as in the view)
  ViewPtrF vf = RING ? (ViewPtrF)(vw + kViewXiF) : (ViewPtrF)st4->xi_f;
  double R[9], t[3];
  float Tf[12];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = vx[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = vx[9 + i];
#pragma unroll
  for (int i = 0; i < 12; ++i) Tf[i] = vf[i];
  const bool do_err = (a.mode & 1) && have_lin;
  const float4* __restrict__ tpt_old = a.tpt[cur];
  const double* __restrict__ mahal_old = a.mahal[cur];
  float4* __restrict__ tpt_new = a.tpt[nxt];
  double* __restrict__ mahal_new = a.mahal[nxt];
  double wave_total = 0.0;
  unsigned int ncand = 0, nvalid = 0, nstaged = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G, grp = lane / G;
  WaveStage& S = stage_all[wave];
  NG_STAMP(0);
  const unsigned long long t_start = a.grp_cost ? __builtin_amdgcn_s_memtime() : 0ull;
#define NG_HAVE_LIN have_lin
#include "ngicp_pass_group.inc"
#undef NG_HAVE_LIN
}

// ---- k_gicp_head: ONE launch per iteration, without a grid-wide meeting.  What the solver launch costs an iteration is not its work
//      (~1.6 us of serial FP64 and one read) but its dispatch and the two kernel boundaries around it (13.6 us of 48 at c3).  Here the
//      kernel boundary between two passes is the only synchronisation: the blocks of pass i add their rows up per SUBSET of groups
//      (subset = group index mod 32, the solver's own summation order: the last block of a subset to finish - a ticket per subset,
//      rows written through, read back with agent-scope loads, no fence - adds its subset's rows in ascending group order), and every
//      block of launch i + 1 begins by reading the state image and those 32 rows, adds them in subset order and runs the optimiser's
//      step ITSELF (lm_step, the very function the solver runs; same inputs, same order, so every block arrives at the same pose, and
//      at the bits the two-launch path arrives at).  Block 0 of every launch is the SOLVER BLOCK: the same step through lm_solve_body,
//      with everything that leaves the kernel - the advanced state image (into the other of two buffers: the launch's own blocks are
//      still reading this one), trace rows, the progress word, the final image for the host, the next launch order (into the other
//      of two order buffers, for the same reason).  One more launch than passes: the head that consumes the last pass's rows. ----
struct HeadShared {
  LmHot L;
  double sums[kPartialStride];
  double crow[kSolveRowSubsets][kNumSlots + 1];
};
__device__ __forceinline__ double uniform_double(double v) {  // (a value every lane holds alike, into scalar registers)
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int G, int WPS>
__global__ void __launch_bounds__(256, WPS) k_gicp_head(PassArgs a) {
  constexpr bool FUSED = true;  // (rows and costs are stored write-through: another block of this launch reads the rows)
  constexpr int B = 64 / G;
  constexpr int kWin = WPS >= 4 ? 12 : 16, kSideStep = WPS >= 4 ? 8 : 16;  // (see k_gicp_pass)
  static_assert(kWin <= kSortedPad && B == kBatchQueries, "see k_gicp_pass");
  __shared__ double lds[4][kNumSlots];
  union HeadPassShared {
    WaveStage stage[4];
    SolveShared<256> sv;  // the solver block
    HeadShared hs;        // a search block's head, before its search tables
  };
  __shared__ HeadPassShared shm;
  __shared__ int last_block;
  // Everything the head needs from memory is requested at once - the done word, the state image, the 32 subset rows - and looked at
  // afterwards: one round trip instead of three in a row.
  const unsigned long long t_entry = a.dbg_stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  const LmState* __restrict__ st = a.st;
  const int n_groups = (int)gridDim.x - 1;
  {
    HeadShared& H = shm.hs;
    constexpr int kHotWords = (int)(sizeof(LmHot) / 4);
    static_assert(kHotWords <= 2 * 256, "two words of the state image per thread");
    static_assert(kSolveRowSubsets * kNumSlots == 256 * 4, "a thread fetches four doubles of the subset rows");
    const int done_before = *a.done_flag;  // (an earlier launch has ended the alignment; the host had enqueued this one ahead)
    const int w0 = threadIdx.x, w1 = threadIdx.x + 256;
    const int i0 = reinterpret_cast<const int*>(&st->hot)[w0], i1 = w1 < kHotWords ? reinterpret_cast<const int*>(&st->hot)[w1] : 0;
    const int c = threadIdx.x >> 3, v0 = (threadIdx.x & 7) * 4;
    const double* src = a.crow_in + c * kNumSlots + v0;
    const double r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];  // (zeros before the first pass)
    if (done_before) return;
    reinterpret_cast<int*>(&H.L)[w0] = i0;
    if (w1 < kHotWords) reinterpret_cast<int*>(&H.L)[w1] = i1;
    H.crow[c][v0] = r0; H.crow[c][v0 + 1] = r1; H.crow[c][v0 + 2] = r2; H.crow[c][v0 + 3] = r3;
  }
  __syncthreads();
  const bool pending = shm.hs.L.pending != 0;  // (block-uniform)
  if (blockIdx.x == 0) {
    // ---- the solver block
    __syncthreads();  // (lm_solve_body lays its scratch over the head's)
    if (pending) {
      lm_solve_body<256, false>(a.sa, shm.sv);  // (a.sa: partials = the 32 subset rows, st_out = the other state buffer)
      __syncthreads();
      if (threadIdx.x == 0 && shm.sv.L.done) *a.done_flag = 1;
    } else {
      // the first launch of an alignment: nothing to consume yet; the state moves on unchanged, marked as having a pass under way
      LmState* sto = a.sa.st_out;
      for (int w = threadIdx.x; w < (int)(sizeof(LmHot) / 4); w += 256) reinterpret_cast<int*>(&sto->hot)[w] = reinterpret_cast<const int*>(&st->hot)[w];
      if (threadIdx.x < 12) sto->xi_f[threadIdx.x] = st->xi_f[threadIdx.x];
      __syncthreads();
      if (threadIdx.x == 0) sto->hot.pending = 1;
    }
    return;
  }
  // ---- head of a search block: the state this pass runs at.  Wave 0 adds the subset rows (lane v: slot v, subsets in order - the
  //      solver's own order) and its lane 0 steps the optimiser; the other waves wait at the barrier.
  if (pending && threadIdx.x < 64) {
    HeadShared& H = shm.hs;
    if (threadIdx.x < kPartialStride) {
      double tsum = 0.0;
      if (threadIdx.x < kNumSlots)
        for (int c = 0; c < kSolveRowSubsets; ++c) tsum += H.crow[c][threadIdx.x];
      H.sums[threadIdx.x] = tsum;
    }
    wave_lds_sync();
    if (threadIdx.x == 0) lm_step(H.L, a.sa.cfg, H.sums, nullptr, 0);  // (no trace row: the solver block writes it)
  }
  __syncthreads();
  if (shm.hs.L.done) return;  // (block-uniform; the solver block tells the host)
  const int have_lin = __builtin_amdgcn_readfirstlane(shm.hs.L.have_lin);
  const int cur = __builtin_amdgcn_readfirstlane(shm.hs.L.cur) & 1, nxt = cur ^ 1;
  double R[9], t[3];
  float Tf[12];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = uniform_double(shm.hs.L.xi.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = uniform_double(shm.hs.L.xi.t[i]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) Tf[r * 4 + c] = (float)R[r * 3 + c];
    Tf[r * 4 + 3] = (float)t[r];
  }
  __syncthreads();  // (the search tables lie where the head's scratch was)
  const bool do_err = (a.mode & 1) && have_lin;
  const bool do_lin = (a.mode & 2);
  const float4* __restrict__ tpt_old = a.tpt[cur];
  const double* __restrict__ mahal_old = a.mahal[cur];
  float4* __restrict__ tpt_new = a.tpt[nxt];
  double* __restrict__ mahal_new = a.mahal[nxt];
  const Grid& g = a.grid;
  double wave_total = 0.0;
  unsigned int ncand = 0, nvalid = 0, nstaged = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G, grp = lane / G;
  WaveStage& S = shm.stage[wave];
  if (a.dbg_stamps && lane == 0) {  // diagnostic only: entry, end of the head
    unsigned long long* d = a.dbg_stamps + (size_t)(blockIdx.x * 4 + wave) * kStampStride;
    d[20] = t_entry;
    d[21] = __builtin_amdgcn_s_memtime();
  }
  NG_STAMP(0);
  const int slot = (int)blockIdx.x - 1;
  const int group = (a.grp_order && a.order_valid && *a.order_valid) ? a.grp_order[slot] : slot;
  const unsigned long long t_start = a.grp_cost ? __builtin_amdgcn_s_memtime() : 0ull;
  if (a.t_first && slot == 0 && threadIdx.x == 0 && !have_lin) *a.t_first = __builtin_amdgcn_s_memrealtime();
#define NG_HAVE_LIN have_lin
#include "ngicp_pass_group.inc"
#undef NG_HAVE_LIN
  // ---- tail: the last block of the group's subset adds the subset's rows
  const int c = group % kSolveRowSubsets;
  if (wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const int members = (n_groups - c + kSolveRowSubsets - 1) / kSolveRowSubsets;
      const int tk = __hip_atomic_fetch_add(a.cluster_ticket + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_block = (tk == members - 1) ? 1 : 0;
    }
  }
  __syncthreads();
  if (a.dbg_stamps && lane == 0) a.dbg_stamps[(size_t)(blockIdx.x * 4 + wave) * kStampStride + 22] = __builtin_amdgcn_s_memtime();  // (ticket drawn)
  if (!last_block || wave != 0) return;
  if (lane < kNumSlots) {
    // rows c, c + 32, c + 64, ... in ascending order, sixteen loads in flight at a time; rows beyond the grid add 0.0 (as the solver does)
    constexpr int kChunk = 16;
    const double* rows = a.partials + lane;
    double acc = 0.0;
    for (int g0 = c; g0 < n_groups; g0 += kSolveRowSubsets * kChunk) {
      double v[kChunk];
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const int gi = g0 + j * kSolveRowSubsets;
        v[j] = gi < n_groups ? __hip_atomic_load(rows + (size_t)gi * kNumSlots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
      if (g0 == c) acc = v[0]; else acc += v[0];
#pragma unroll
      for (int j = 1; j < kChunk; ++j) acc += v[j];
    }
    a.crow_out[c * kNumSlots + lane] = acc;
  }
  if (lane == 0) __hip_atomic_store(a.cluster_ticket + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch: ordered by the kernel boundary)
  if (a.dbg_stamps && lane == 0) a.dbg_stamps[(size_t)(blockIdx.x * 4 + wave) * kStampStride + 23] = __builtin_amdgcn_s_memtime();  // (subset row stored)
}

// ---- the persistent kernel: ONE launch per alignment.  A thin loop over passes and over the block's groups around two CALLED functions
//      (the group body above, the solver): what the loop keeps alive does not compete with the search for registers.  The functions
//      find the kernel's arguments where the hardware put them (the kernel-argument segment), and share the block's LDS as
//      namespace-scope variables (allocated to this kernel only). ----
union PersistShared {
  WaveStage stage[4];
  SolveShared<256> sv;  // the solver (the last block to arrive) works where the search tables were
};
__shared__ PersistShared g_persist_shm;
__shared__ double g_persist_lds[4][kNumSlots];
__shared__ int g_persist_last;
typedef const PassArgs __attribute__((address_space(4))) KernelPassArgs;

// (ka: the kernel's argument segment, handed down by the kernel itself - inside a called function __builtin_amdgcn_kernarg_segment_ptr()
// is the pointer to the kernel's IMPLICIT arguments, behind the explicit ones)
// (Arguments of a called function arrive in vector registers: made wave-uniform again with v_readfirstlane, so that a.field is a
// scalar load and group / pass_no live in scalar registers.)
__device__ __forceinline__ KernelPassArgs* uniform_args(KernelPassArgs* ka) {
  const unsigned long long v = (unsigned long long)ka;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return (KernelPassArgs*)(((unsigned long long)hi << 32) | lo);
}
template <int G, int WPS>
__device__ __attribute__((noinline)) void persist_group_call(KernelPassArgs* ka, int group, int pass_no) {
  persist_group_body<G, WPS, true>(*uniform_args(ka), __builtin_amdgcn_readfirstlane(group), __builtin_amdgcn_readfirstlane(pass_no), g_persist_shm.stage, g_persist_lds);
}

__device__ __attribute__((noinline)) void persist_solve_call(KernelPassArgs* ka) {
  const char* kargs = (const char*)uniform_args(ka);
  lm_solve_body<256, true, true>(*reinterpret_cast<const SolveArgs*>(kargs + offsetof(PassArgs, sa)), g_persist_shm.sv);
}

// ---- k_gicp_queue: one launch per pass, as k_gicp_pass, but a grid of as many blocks as are RESIDENT together, each drawing its next
//      group from a counter (launch order = the order of the draws: heaviest first as before) instead of leaving a freed slot to the
//      dispatcher, which does not refill it at once (c5, 2217 groups over 1024 slots: the slots 15-20 % empty between rounds).  No
//      block waits for another; the solver is a launch of its own as ever. ----
template <int G, int WPS>
__device__ __attribute__((noinline)) void queue_group_call(KernelPassArgs* ka, int group) {
  persist_group_body<G, WPS, false>(*uniform_args(ka), __builtin_amdgcn_readfirstlane(group), 0, g_persist_shm.stage, g_persist_lds);
}

template <int G, int WPS>
__global__ void __launch_bounds__(256, WPS) k_gicp_queue(PassArgs a) {
  if (!(a.mode & 4) && a.st->hot.done) return;
  KernelPassArgs* ka = (KernelPassArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const bool ordered = a.grp_order && a.order_valid && *a.order_valid;
  const int n_groups = (a.n_batches + 3) / 4;
  // The first group of a block is the one of its own number (a thousand blocks drawing from ONE word at the same moment queue up at the
  // memory channel that holds it: measured +10 us per pass at c3, +28 at c5); from then on position gridDim.x + draw.
  int slot = (int)blockIdx.x;
  for (;;) {
    if (a.t_first && slot == 0 && threadIdx.x == 0 && !a.st->hot.have_lin) *a.t_first = __builtin_amdgcn_s_memrealtime();
    queue_group_call<G, WPS>(ka, ordered ? a.grp_order[slot] : slot);
    if (threadIdx.x == 0) g_persist_last = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int draw = g_persist_last;
    __syncthreads();  // (the word is written again in the next turn; the group's row of sums in LDS has been read)
    slot = (int)gridDim.x + draw;
    if (slot >= n_groups) {
      // every block draws exactly one number beyond the groups - n_groups draws in all; the last puts the counter back for the next launch
      if (draw == n_groups - 1 && threadIdx.x == 0) __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

template <int G, int WPS>
__global__ void __launch_bounds__(256, WPS) k_gicp_persist(PassArgs a) {
  KernelPassArgs* ka = (KernelPassArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // == &a: where the hardware put the arguments
  const bool ordered = a.grp_order && a.order_valid && *a.order_valid;  // (fixed for the whole alignment: the solver builds the NEXT alignment's order)
  const int n_groups = (a.n_batches + 3) / 4;
  for (int pass_no = 0; pass_no < a.max_passes; ++pass_no) {
    {
      // (the address is made opaque HERE, behind the wait that released the pass: no load of the entry can be placed earlier)
      const int* vw_g = a.st->view + (size_t)(a.first_pass + pass_no) * kViewWords;
      asm volatile("" : "+s"(vw_g));
      typedef const int __attribute__((address_space(4))) * ViewPtr;
      ViewPtr vw = (ViewPtr)(unsigned long long)vw_g;
      if (vw[kViewDone]) break;
      if (a.t_first && blockIdx.x == 0 && threadIdx.x == 0 && !vw[kViewHaveLin])
        __hip_atomic_store(a.t_first, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The grid is as many blocks as are resident together; block b takes the positions b, 2P - 1 - b, 2P + b, ... of the launch order
    // (heaviest first, then back and forth), the same ones in every pass: a group's correspondences and Mahalanobis matrices stay
    // in its block's CU / XCD from pass to pass.
    for (int turn = 0;; ++turn) {
      const int slot = turn * (int)gridDim.x + ((turn & 1) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x);
      if (slot >= n_groups) break;
      __syncthreads();  // (the block's row of sums in LDS has been read)
      persist_group_call<G, WPS>(ka, ordered ? a.grp_order[slot] : slot, pass_no);
    }
    // ---- the blocks meet here after every pass.  Every store of this block that another block will read was made by wave 0 as a
    //      write-through store; wave 0 drains them, then one lane takes a ticket (relaxed, agent scope; the ticket counts on, pass after
    //      pass).  The last block to arrive steps the optimiser (no fence: lm_solve_body<.., PERSIST> reads what other blocks wrote with
    //      agent-scope loads and stores the state write-through), then releases the next pass; the others wait for that with one
    //      polling lane each.  Every block of the grid is resident (the host sizes the grid by the occupancy of this very kernel and
    //      launches it cooperatively), the wait is bounded all the same: a block that gives up raises `gen` to a value no pass reaches,
    //      so that every other block leaves too and the grid drains. ----
    constexpr int kGenFailed = 1 << 30;
    if (threadIdx.x < 64) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) {
        const int tk = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g_persist_last = (tk == (pass_no + 1) * (int)gridDim.x - 1) ? 1 : 0;
      }
    }
    __syncthreads();
    static_assert(kGenLines == 256, "one thread of the releasing block per copy of the release word");
    if (g_persist_last) {  // (block-uniform)
      if (a.sa.pass_ticks && threadIdx.x == 0) a.sa.pass_ticks[2 * pass_no] = __builtin_amdgcn_s_memrealtime();
      persist_solve_call(ka);
      if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the state image and the next view were stored by wave 0: its own counter covers them)
      __syncthreads();
      if (a.sa.pass_ticks && threadIdx.x == 0) a.sa.pass_ticks[2 * pass_no + 1] = __builtin_amdgcn_s_memrealtime();
      __hip_atomic_fetch_max(a.gen + threadIdx.x * kGenStride, pass_no + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) g_persist_last = 0;
    } else if (a.persist == 2) {
      return;  // (A/B measurement only, one pass per launch: nobody waits)
    } else if (threadIdx.x == 0) {
      // (every poll goes out to memory, and the blocks still working feel ~770 pollers: c3 41 us per pass at one poll per microsecond
      // and block, 36-37 at one per 3.4 us - s_sleep's maximum; sleeping without polls for 3/4 of the previous period first: no better)
      const int* my_gen = a.gen + ((int)blockIdx.x % kGenLines) * kGenStride;
      int seen = 0;
      for (int spin = 0; spin < (1 << 20); ++spin) {  // (~5 s)
        seen = __hip_atomic_load(my_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen > pass_no) break;
        __builtin_amdgcn_s_sleep(127);
      }
      g_persist_last = (seen <= pass_no || seen >= kGenFailed) ? 2 : 0;
    }
    __syncthreads();
    if (g_persist_last == 2) {  // (block-uniform; the host sees no done flag and falls back to one launch per pass)
      __hip_atomic_fetch_max(a.gen + threadIdx.x * kGenStride, kGenFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __syncthreads();  // (g_persist_last is written again in the next pass)
  }
}
#undef NG_STAMP


}  // namespace ngk
#include "ngicp_pass_st.h"
namespace ngk {

// map correspondences (sorted source slot -> sorted target position) back to ORIGINAL indices
__global__ void __launch_bounds__(256) k_corr_to_original(const float4* __restrict__ tpt, const float4* __restrict__ qpts, const float4* __restrict__ src_sorted,
                                                           const float4* __restrict__ tgt_sorted, int n, int* __restrict__ out_corr, float* __restrict__ out_sqd,
                                                           const float* __restrict__ xi_f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 sp = qpts[i];
  const int o = __float_as_int(src_sorted[__float_as_int(sp.w)].w);
  const int j = __float_as_int(tpt[i].w);
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
