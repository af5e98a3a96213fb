// C-ABI implementation (include/ngicp.h) — host driver for the HIP kernels.
// One handle == one nano_gicp::NanoGICP instance (/root/reference/include/nano_gicp/nano_gicp.hpp:58-137).
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
#include "../../include/ngicp.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <sched.h>
#include <string>
#include <vector>

#include "ngicp_pass.h"
#include "ngicp_cloudops.h"
#include "ngicp_filters.h"

using namespace ngk;

namespace {

thread_local std::string g_create_error;

struct HipError {
  hipError_t code;
  const char* what;
  const char* file;
  int line;
};

#define HIP_TRY(expr)                                          \
  do {                                                         \
    hipError_t _e = (expr);                                    \
    if (_e != hipSuccess) throw HipError{_e, #expr, __FILE__, __LINE__}; \
  } while (0)

struct ArgError {
  int code;
  std::string msg;
};

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

std::atomic<long long> g_device_allocs{0};  // hipMalloc calls of the engine's buffers (ngicp_stats::device_allocs)

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&p, want));
    g_device_allocs.fetch_add(1, std::memory_order_relaxed);
    cap = want;
  }
  bool ensure_grew(size_t bytes) {  // true when the buffer was (re)allocated: its contents are gone
    const void* before = p;
    ensure(bytes);
    return p != before;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// An uploaded, cell-sorted, indexed cloud.  Shared between handles (odom.cc:525) and between the
// source/target slots (swapSourceAndTarget) through shared_ptr.
struct DeviceCloud {
  size_t n = 0;
  DevBuf sorted;      // float4[kSortedPad + n + kSortedPad]: the cell-sorted points between two runs of far-away sentinels, so
                      // that the 8-point windows of the search may overhang the array's ends without index clamps
  float4* pts() const { return sorted.as<float4>() + kSortedPad; }
  DevBuf sorted3;     // Xyz[kSortedPad + n + kSortedPad]: the same points and sentinels, 12 bytes each (the pass kernel's walks)
  Xyz* xyz3() const { return sorted3.as<Xyz>() + kSortedPad; }
  DevBuf sortedp;     // float4[kSortedPad + n + kSortedPad]: the same points and sentinels, w = sorted position (what the staged pass copies to LDS)
  float4* xyzp() const { return sortedp.as<float4>() + kSortedPad; }
  DevBuf perm;        // int[n]    sorted position -> original index
  DevBuf inv_perm;    // int[n]    original index -> sorted position (lazily built)
  bool has_inv = false;
  DevBuf cell_start;  // int[kCellPad + ncells + 1 + kCellPad]: the exclusive prefix of points per cell, framed by kCellPad entries on each side (0 in
                      // front, n behind) so that the pass may fetch the four bounds around a cell with ONE 16-byte load at any cell
  int* cells() const { return cell_start.as<int>() + kCellPad; }
  DevBuf cell_box;    // uint[kCellPad + ncells + kCellPad]: the (y,z) extent of every cell's points inside the cell (k_cell_boxes), framed by
                      // empty boxes; only when the building handle had NGICP_CELL_BOXES on
  bool has_boxes = false;
  DevBuf qpts;        // float4[n]  the points in Morton-tile query order, w = sorted position
  DevBuf batches;     // int2[n_batches] {first qpts index, count <= 32}: tile-aligned query batches
  DevBuf n_batches_dev;
  DevBuf batch_boxes; // float[n_batches][6] centre + half extents of each batch (cloud frame)
  int n_batches = 0;
  Grid grid{};
  float bb_min[3] = {0.f, 0.f, 0.f}, bb_max[3] = {0.f, 0.f, 0.f};  // bounding box of the points (a submap's box is the union of its keyframes')
  double build_ms = 0.0;
  int device = 0;
};

// Index objects are recycled: a LiDAR pipeline builds a new source index per scan, and hipMalloc / hipFree of its nine
// buffers (both synchronise the device) cost more than building the index.  The last owner hands the object back to a
// per-process pool; a build takes one of the right device from it and only grows the buffers that are too small.
struct CloudPool {
  std::mutex m;
  std::vector<DeviceCloud*> free_list;
};
CloudPool& cloud_pool() {
  static CloudPool* p = new CloudPool;  // never destroyed: device memory must not be freed after the HIP runtime has shut down
  return *p;
}
// the same for covariance sets (one per scan, 48 bytes per point)
struct BufPool {
  std::mutex m;
  std::vector<std::pair<int, DevBuf*>> free_list;  // {device, buffer}
};
BufPool& buf_pool() {
  static BufPool* p = new BufPool;
  return *p;
}
// A recycled object may still be read by work that ANOTHER handle has in flight (a shared source index, a keyframe's covariance
// set).  Instead of a device-wide, host-blocking hipDeviceSynchronize() the acquiring handle's stream waits - on the device - for
// what every live handle of the same GPU has enqueued so far: one event record + one stream wait per handle (DLO has two).
struct HandleRegistry {
  std::mutex m;
  std::vector<ngicp*> live;
};
HandleRegistry& registry() {
  static HandleRegistry* r = new HandleRegistry;
  return *r;
}
void fence_engine_streams(ngicp* h);  // defined below struct ngicp

std::shared_ptr<DevBuf> acquire_buf(ngicp* h, int device, size_t bytes) {
  DevBuf* b = nullptr;
  {
    BufPool& bp = buf_pool();
    std::lock_guard<std::mutex> lock(bp.m);
    size_t best = bp.free_list.size();
    for (size_t i = 0; i < bp.free_list.size(); ++i)  // best fit: a scan's set must not take the submap's buffer
      if (bp.free_list[i].first == device && bp.free_list[i].second->cap >= bytes &&
          (best == bp.free_list.size() || bp.free_list[i].second->cap < bp.free_list[best].second->cap))
        best = i;
    if (best < bp.free_list.size()) {
      b = bp.free_list[best].second;
      bp.free_list.erase(bp.free_list.begin() + (long)best);
    }
  }
  if (b) {
    fence_engine_streams(h);  // previous owners' work on other streams
  } else {
    b = new DevBuf;
    b->ensure(bytes);
  }
  return std::shared_ptr<DevBuf>(b, [device](DevBuf* p) {
    BufPool& bp = buf_pool();
    std::lock_guard<std::mutex> lock(bp.m);
    if (bp.free_list.size() < 12) bp.free_list.emplace_back(device, p);
    else delete p;
  });
}

std::shared_ptr<DeviceCloud> acquire_cloud(ngicp* h, int device) {
  DeviceCloud* dc = nullptr;
  {
    CloudPool& cp = cloud_pool();
    std::lock_guard<std::mutex> lock(cp.m);
    for (size_t i = 0; i < cp.free_list.size(); ++i)
      if (cp.free_list[i]->device == device) {
        dc = cp.free_list[i];
        cp.free_list.erase(cp.free_list.begin() + (long)i);
        break;
      }
  }
  if (dc) {
    // its previous owners may still have work in flight on their streams that reads the buffers
    fence_engine_streams(h);
    dc->n = 0;
    dc->has_inv = false;
    dc->n_batches = 0;
    dc->build_ms = 0.0;
  } else {
    dc = new DeviceCloud;
    dc->device = device;
  }
  return std::shared_ptr<DeviceCloud>(dc, [](DeviceCloud* p) {
    CloudPool& cp = cloud_pool();
    std::lock_guard<std::mutex> lock(cp.m);
    if (cp.free_list.size() < 8) cp.free_list.push_back(p);
    else delete p;
  });
}

// Covariances, packed symmetric FP64 [n][6], stored in the sorted order of `order`.
struct CovSet {
  std::shared_ptr<DevBuf> data;
  size_t n = 0;
  std::shared_ptr<DeviceCloud> order;
  void clear() {
    data.reset();
    order.reset();
    n = 0;
  }
};

struct Slot {
  std::shared_ptr<DeviceCloud> dev;
  const float* host = nullptr;  // pending (registered, not yet uploaded) cloud
  size_t n = 0;
  size_t stride = 0;
  uint64_t identity = 0;
  bool present = false;
  void clear() {
    dev.reset();
    host = nullptr;
    n = stride = 0;
    identity = 0;
    present = false;
  }
};

struct LoopCtx;  // the per-alignment launch arguments (defined with the registration loop)
constexpr int kMaxPersistPasses = 2048;  // alignments with more possible passes than this take one launch per pass (the ring of views is 512 bytes per pass)
constexpr int kMaxTickPasses = 1024;  // passes of an alignment whose timestamps the persistent kernel records when profiling is on
constexpr int kShardSlots = 4, kShardLag = 2;  // point-sharded stepping: the `done` word of step k is read at step k + kShardLag

struct Params {
  int k = 20;                                                       // impl/nano_gicp_impl.hpp:57
  double max_corr_dist = (double)std::numeric_limits<float>::max(); // :59
  int max_iter = 64;                                                // impl/lsq_registration_impl.hpp:52
  double trans_eps = 5e-4;                                          // :54
  double rot_eps = 2e-3;                                            // :53
  int optimizer = NGICP_OPT_LEVENBERG_MARQUARDT;                    // :56
  int lm_max_iter = 10;                                             // :58
  double lm_init_lambda_factor = 1e-9;                              // :59
  int regularization = NGICP_REG_PLANE;                             // impl/nano_gicp_impl.hpp:61
  int num_threads = 0;
};

}  // namespace

struct ngicp {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  hipEvent_t ev_cov_a = nullptr, ev_cov_b = nullptr;  // around the last covariance kernel; read lazily (ngicp_get_stats)
  hipEvent_t ev_fence = nullptr;                       // fence_engine_streams()
  bool cov_timing_pending = false;
  std::string err;
  Params p;
  double voxel_size = 0.0;  // 0 = auto
  double target_occupancy = 24.0;  // mean points a random point sees in its own cell; tuned on MI355X (c2/c3/c5 workloads)
  int host_wait = 0;        // 0: poll without giving the core up, 1: sched_yield between polls (ngicp_set_host_wait)
  CovSet shard_covs[2];     // covariance sets being computed in blocks by several ranks (ngicp_covs_shard_*), uncommitted
  int chunk_pairs = 3;      // (pass, solve) pairs kept in flight ahead of the solver's published progress (env NGICP_CHUNK)
  int stage_grow = 6;       // upper limit of rings served from the LDS stage
  // {cloud size, auto voxel edge} of recent builds, one entry per size class (a factor of two around the size): a DLO pipeline has three
  // or four - the scan, the voxel-filtered keyframe made from it, the submap - and each would otherwise pay the refinement passes again
  std::pair<size_t, double> voxel_memo[4] = {{0, 0.0}, {0, 0.0}, {0, 0.0}, {0, 0.0}};
  int voxel_memo_next = 0;
  bool profiling = false;
  int prof_stride = 1;      // time every prof_stride-th pass launch (events between kernels cost a few microseconds each)

  Slot src, tgt;
  CovSet src_covs, tgt_covs;

  // workspaces
  DevBuf raw, unsorted, keys, counts, fill, tile_sums, tile_sq, tmp, bbox, occ;
  int pass_slots = 768;  // blocks of the 3-waves-per-SIMD pass kernel resident on this device at once
  int persist_slots = 0; // blocks of the persistent pass kernel resident at once (its grid), 0: not available
  int queue_slots[2] = {768, 1024};  // blocks of k_gicp_queue<2, 3> / <2, 4> resident at once
  int persist = 0;       // env NGICP_PERSIST=1: ONE launch per alignment (k_gicp_persist).  Exact and complete, but measured no faster than one
                         // launch per pass: off by default
  int order_sel = 0;     // which of the two launch-order buffers (and flag words) the next alignment reads
  DevBuf grp_order_alt;  // the second order buffer: the persistent kernel's solver builds the NEXT alignment's order there
  DevBuf gen_lines;      // the persistent kernel's release word, kGenLines copies (PassArgs::gen)
  int cell_boxes = 0;    // env NGICP_CELL_BOXES=1: per-cell (y,z) extents built with every index and used by the pass (see "Index build")
  int head = 0;          // env NGICP_HEAD=1: k_gicp_head - no solver launch, every block steps the optimiser at its head
  DevBuf state_alt;      // k_gicp_head: the second state buffer (a launch's solver block writes the one its blocks are not reading)
  DevBuf head_ws;        // k_gicp_head: {done flag (64 B), subset tickets (128 B), subset rows of even / odd launches (2 x 8 KB)}
  unsigned long long* pin_ticks = nullptr;  // pinned [2 * kMaxTickPasses]: per pass {last block arrived, next pass released} (profiling)
  double prev_staged_fraction = -1.0;  // share of the queries the previous alignment served through row lists (-1: none yet)
  DevBuf dbg, dbg_q, dbg_s, dbg_span, grp_order, grp_cost, batch_far;
  const void* order_src = nullptr;  // source index / group count the contents of grp_order were built for
  int order_groups = -1;
  DevBuf tpt[2], mahal[2], partials, state, trace, tfinal, out_xyz, scratch16, queries, knn_idx, knn_d2, sums, ticket;
  std::vector<hipEvent_t> prof_events;  // pairs around each pass launch when profiling is on
  int* h_progress = nullptr;  // pinned: {passes done | kProgressDone}, written by the solver (SolveArgs::progress_host)
  LmState* pin_state = nullptr;  // pinned [2]: the state image an align uploads / the one it reads back (no staging copies)
  LmHot* pin_final = nullptr;    // pinned: the state image the solver writes when an alignment is done (SolveArgs::final_host)
  DevBuf order_flag, t_first;    // device words {order flag 0, ticket, gen, order flag 1}: grp_order / grp_order_alt holds a complete order, the
                                 // fused / persistent kernels' ticket and released-pass counter; 100 MHz stamp of the alignment's first pass
  int hook_valid = 0;     // 1: the linearize hook has produced correspondences; 2: an align has (indices of its last linearisation)

  // results of the last align
  float final_T[16];
  double final_hessian[36];
  int converged = 0, nr_iterations = 0;
  std::vector<double> trace_host;
  size_t trace_rows_dev = 0;  // rows of the last align's LM trace still on the device
  ngicp_stats stats{};

  // sharded stepping
  bool sharded_active = false;
  std::shared_ptr<LoopCtx> shard_ctx;      // the loop context of the alignment being stepped (one prepare_loop per alignment)
  hipEvent_t ev_shard[kShardSlots] = {};   // behind the copy of the `done` word of step k (slot k mod kShardSlots)
  int* h_shard_done = nullptr;             // pinned [kShardSlots]
  long shard_steps = 0;
  hipStream_t shard_stream = nullptr;      // the stream the last step was enqueued on

  // scan preprocessing / map voxel filter
  FilterWorkspace fws;
  DevBuf xyzi, map_pts;      // the unpacked input of a filter call; the accumulated map, float4 {x, y, z, intensity}
  size_t map_n = 0;
  const float4* filt_out = nullptr;  // result of the last preprocess call (device memory of fws / xyzi), filt_n points
  int filt_n = 0;

  // device-resident keyframe store (src/dlo/odom.cc keyframes + keyframe_normals) and the submap assembled from it
  struct Keyframe {
    std::shared_ptr<DeviceCloud> cloud;  // indexed, cell-sorted
    std::shared_ptr<DevBuf> covs;        // [n][6] FP64 in the cloud's sorted order
  };
  std::vector<Keyframe> keyframes;
  std::vector<int> submap_ids;           // keyframes of the submap that is the current target (valid while submap_cloud is the target)
  const DeviceCloud* submap_cloud = nullptr;
};

namespace {

void fence_engine_streams(ngicp* h) {
  HandleRegistry& r = registry();
  std::lock_guard<std::mutex> lock(r.m);
  for (ngicp* o : r.live) {
    if (o == h || o->device != h->device) continue;  // (work on h's own stream is ordered before anything h enqueues next)
    HIP_TRY(hipEventRecord(h->ev_fence, o->stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_fence, 0));
  }
}

int pick_blocks(size_t work_items, int per_block, int max_blocks) {
  size_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (size_t)max_blocks) b = max_blocks;
  return (int)b;
}

// ------------------------------------------------------------------------------------------
// Index build
// ------------------------------------------------------------------------------------------
// NGICP_CELL_BOXES=1 (experiment, round 3): per-cell (y,z) extents of the points, built with every index; the pass cuts and prunes its
// ring-1 (query, row) pairs with them.  Exact, a quarter fewer candidates (c3 68.4 -> 52.2 per query, c5 25.9 -> 23.8), 25 % fewer
// ring-1 units per wave - and the launch exactly as long as before (c3 33.7 us either way, c5 41.3 -> 42.5: the second load per row):
// the units it removes were served in parallel by lanes that would otherwise idle.  Off by default.
// (read at ngicp_create: ngicp::cell_boxes; an index carries boxes when the handle that built it had the switch on)

Grid make_grid(const float mn[3], const float mx[3], double h, int max_cells) {
  Grid g{};
  double ext[3];
  for (int d = 0; d < 3; ++d) ext[d] = std::max(0.0, (double)mx[d] - (double)mn[d]);
  for (;;) {
    double nx = std::floor(ext[0] / h) + 1, ny = std::floor(ext[1] / h) + 1, nz = std::floor(ext[2] / h) + 1;
    // (the pass kernel packs a listed row as y | z << 16 in one int: y < 65536, z < 32768)
    if (nx * ny * nz <= (double)max_cells && ny < 65536.0 && nz < 32768.0) {
      g.nx = (int)nx;
      g.ny = (int)ny;
      g.nz = (int)nz;
      break;
    }
    h *= 1.26;
  }
  g.ox = mn[0];
  g.oy = mn[1];
  g.oz = mn[2];
  g.h = (float)h;
  g.inv_h = 1.0f / g.h;
  g.ncells = g.nx * g.ny * g.nz;
  double span = std::max(ext[0], std::max(ext[1], ext[2]));
  g.slack = (float)(1e-3 * h + 4e-6 * (span + std::fabs(mn[0]) + std::fabs(mn[1]) + std::fabs(mn[2])));
  return g;
}

// occ_host: where to put sum(count^2) (synchronous read-back), or null.  occ_device_only: compute it into h->occ but leave it
// on the device (the caller reads it later, when it synchronises anyway).
// stride > 1: an occupancy ESTIMATE from every stride-th point (the cell-start table written is then meaningless)
void count_and_scan(ngicp* h, DeviceCloud& dc, int n, unsigned long long* occ_host, bool occ_device_only = false, int stride = 1) {
  const Grid& g = dc.grid;
  h->counts.ensure((size_t)(g.ncells + 1) * sizeof(int));
  HIP_TRY(hipMemsetAsync(h->counts.p, 0, (size_t)(g.ncells + 1) * sizeof(int), h->stream));
  hipLaunchKernelGGL(k_cell_count, dim3(pick_blocks((size_t)(n / stride + 1), 256, 2048)), dim3(256), 0, h->stream, h->unsorted.as<float4>(), n, g, h->keys.as<int>(), h->counts.as<int>(),
                     h->fill.as<int>(), stride);
  const int ntiles = (g.ncells + kScanTile - 1) / kScanTile;
  h->tile_sums.ensure((size_t)ntiles * sizeof(int));
  h->tile_sq.ensure((size_t)ntiles * sizeof(unsigned long long));
  dc.cell_start.ensure((size_t)(g.ncells + 1 + 2 * kCellPad) * sizeof(int));
  HIP_TRY(hipMemsetAsync(dc.cell_start.p, 0, kCellPad * sizeof(int), h->stream));  // front pad (k_scan_apply writes the back pad)
  unsigned long long* tsq = (occ_host || occ_device_only) ? h->tile_sq.as<unsigned long long>() : nullptr;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(kScanBlock), 0, h->stream, h->counts.as<int>(), g.ncells, h->tile_sums.as<int>(), tsq);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kScanBlock), 0, h->stream, h->tile_sums.as<int>(), ntiles, (const unsigned long long*)tsq,
                     h->occ.as<unsigned long long>());
  hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(kScanBlock), 0, h->stream, h->counts.as<int>(), g.ncells, h->tile_sums.as<int>(), dc.cells());
  if (occ_host) {
    HIP_TRY(hipMemcpyAsync(occ_host, h->occ.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
}

// Host cloud -> h->unsorted (float4 {x, y, z, bitcast(original index)}) + its bounding box.  The only host read-back of the build
// (the grid dimensions size the allocations).
void stage_host_cloud(ngicp* h, const float* xyz, size_t n, size_t stride, float mn[3], float mx[3]) {
  if (n == 0) throw ArgError{NGICP_ERR_ARG, "empty cloud"};
  if (n > (size_t)0x7fffff00) throw ArgError{NGICP_ERR_ARG, "cloud too large for int indices"};
  if (stride < 12 || (stride % 4) != 0) throw ArgError{NGICP_ERR_ARG, "stride_bytes must be a multiple of 4 and >= 12"};
  const int ni = (int)n;
  const double t0 = now_ms();
  const size_t raw_bytes = (n - 1) * stride + 12;
  h->raw.ensure(raw_bytes);
  HIP_TRY(hipMemcpyAsync(h->raw.p, xyz, raw_bytes, hipMemcpyHostToDevice, h->stream));
  h->unsorted.ensure(n * sizeof(float4));
  const int bbox_blocks = pick_blocks(n, 1024, 512);
  h->bbox.ensure((size_t)bbox_blocks * 8 * sizeof(float));
  hipLaunchKernelGGL(k_unpack_bbox, dim3(bbox_blocks), dim3(256), 0, h->stream, h->raw.as<unsigned char>(), stride, ni, h->unsorted.as<float4>(), h->bbox.as<float>());
  std::vector<float> bb((size_t)bbox_blocks * 8);
  HIP_TRY(hipMemcpyAsync(bb.data(), h->bbox.p, bb.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stats.upload_ms = now_ms() - t0;  // copy + unpack + bounding box (one synchronisation)
  for (int d = 0; d < 3; ++d) mn[d] = 3.0e38f, mx[d] = -3.0e38f;
  float bad = 0.f;
  for (int b = 0; b < bbox_blocks; ++b) {
    for (int d = 0; d < 3; ++d) {
      mn[d] = std::min(mn[d], bb[(size_t)b * 8 + d]);
      mx[d] = std::max(mx[d], bb[(size_t)b * 8 + 3 + d]);
    }
    bad += bb[(size_t)b * 8 + 6];
  }
  // the reference assumes NaNs were removed upstream (src/dlo/odom.cc:443-447); make it an explicit error
  if (bad > 0.f) throw ArgError{NGICP_ERR_ARG, "cloud contains non-finite coordinates"};
  for (int d = 0; d < 3; ++d)
    if (!std::isfinite(mn[d]) || !std::isfinite(mx[d]) || mn[d] > mx[d]) throw ArgError{NGICP_ERR_ARG, "cloud contains non-finite coordinates"};
}

// h->unsorted[0..n) -> an indexed DeviceCloud.  Everything stays on the device; one synchronisation at the end.
std::shared_ptr<DeviceCloud> index_unsorted(ngicp* h, size_t n, const float mn[3], const float mx[3]) {
  auto dc = acquire_cloud(h, h->device);
  dc->n = n;
  for (int d = 0; d < 3; ++d) dc->bb_min[d] = mn[d], dc->bb_max[d] = mx[d];
  const int ni = (int)n;
  HIP_TRY(hipEventRecord(h->ev_a, h->stream));
  h->keys.ensure(n * sizeof(int));
  h->tmp.ensure(n * sizeof(float4));
  h->occ.ensure(2 * sizeof(unsigned long long));
  const int max_cells = 1 << 25;
  double hh;
  const bool auto_h = !(h->voxel_size > 0.0);
  bool memo_hit = false;
  if (!auto_h) {
    hh = h->voxel_size;
  } else {
    double vol = 1.0;
    for (int d = 0; d < 3; ++d) vol *= std::max(0.05, (double)mx[d] - (double)mn[d]);
    hh = std::cbrt(vol / (double)n) * 1.5;  // first guess; refined from measured occupancy below
    hh = std::max(hh, 0.02);
    // consecutive scans / submaps look alike: start from the voxel the last cloud of similar size ended with
    // (saves the refinement passes, each a histogram + scan + host read-back)
    for (const auto& e : h->voxel_memo)
      if (e.second > 0.0 && (double)n > 0.5 * (double)e.first && (double)n < 2.0 * (double)e.first) {
        hh = e.second;
        memo_hit = true;
      }
  }
  dc->grid = make_grid(mn, mx, hh, max_cells);
  unsigned long long occ = 0;
  h->fill.ensure(n * sizeof(int));  // (arrival rank of every point inside its cell: k_cell_count -> k_cell_scatter)
  // with a memoised voxel the occupancy is only checked AFTER the build (it corrects the memo for the next cloud): the
  // read-back then rides on the build's final synchronisation instead of stalling the pipeline here
  count_and_scan(h, *dc, ni, (auto_h && !memo_hit) ? &occ : nullptr, auto_h && memo_hit);
  if (auto_h && !memo_hit) {
    // the first cloud of a size class: refine the voxel edge until the mean occupancy seen by a random point (sum c^2 / n) is near
    // the target.  (Round 3 tried estimating it from every 8th / 32nd point: consecutive points of a LiDAR ring share their cells, a
    // strided sample is not a thinned copy of the cloud, and the estimate settled on cells 2.4x too small - measured, withdrawn.)
    for (int it = 0; it < 4; ++it) {
      const double lam = (double)occ / (double)n;
      const double ratio = h->target_occupancy / std::max(lam, 1.0);
      if (ratio > 0.75 && ratio < 1.33) break;
      double scale = std::pow(ratio, 1.0 / 1.5);
      scale = std::min(4.0, std::max(0.25, scale));
      hh = std::max(0.01, (double)dc->grid.h * scale);
      dc->grid = make_grid(mn, mx, hh, max_cells);
      count_and_scan(h, *dc, ni, &occ);
    }
  }
  const Grid& g = dc->grid;
  dc->sorted.ensure((n + 2 * kSortedPad) * sizeof(float4));
  hipLaunchKernelGGL(k_fill_sentinels, dim3(1), dim3(2 * kSortedPad), 0, h->stream, dc->sorted.as<float4>(), ni);
  dc->perm.ensure(n * sizeof(int));
  hipLaunchKernelGGL(k_cell_scatter, dim3(pick_blocks(n, 256, 2048)), dim3(256), 0, h->stream, h->unsorted.as<float4>(), h->keys.as<int>(), h->fill.as<int>(), ni, dc->cells(),
                     h->tmp.as<float4>());
  hipLaunchKernelGGL(k_cell_rank, dim3(pick_blocks(n, 256, 4096)), dim3(256), 0, h->stream, h->tmp.as<float4>(), ni, g, dc->cells(), dc->pts(),
                     dc->perm.as<int>());
  dc->has_boxes = h->cell_boxes != 0;
  if (dc->has_boxes) {
  dc->cell_box.ensure((size_t)(g.ncells + 2 * kCellPad) * sizeof(unsigned int));
  hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(2 * kCellPad), 0, h->stream, dc->cell_box.as<unsigned int>(), dc->cell_box.as<unsigned int>() + kCellPad + g.ncells, kCellPad, kCellBoxEmpty);
  hipLaunchKernelGGL(k_cell_boxes, dim3((unsigned)((g.ncells + 255) / 256)), dim3(256), 0, h->stream, dc->pts(), dc->cells(), g, dc->cell_box.as<unsigned int>() + kCellPad);
  }
  dc->sorted3.ensure((n + 2 * kSortedPad) * sizeof(Xyz));
  hipLaunchKernelGGL(k_pack_xyz, dim3((unsigned)((n + 2 * kSortedPad + 255) / 256)), dim3(256), 0, h->stream, dc->sorted.as<float4>(), ni + 2 * kSortedPad, dc->sorted3.as<Xyz>());
  dc->sortedp.ensure((n + 2 * kSortedPad) * sizeof(float4));
  hipLaunchKernelGGL(k_pack_pos, dim3((unsigned)((n + 2 * kSortedPad + 255) / 256)), dim3(256), 0, h->stream, dc->sorted.as<float4>(), ni + 2 * kSortedPad, kSortedPad, dc->sortedp.as<float4>());
  {
    // query order (Morton over tiles of 2^shift cells; <= 128 tiles per axis => <= 2M histogram bins)
    int shift = 2;
    while (((std::max(g.nx, std::max(g.ny, g.nz)) - 1) >> shift) >= 128) ++shift;
    int bits = 1;
    while ((1 << bits) <= ((std::max(g.nx, std::max(g.ny, g.nz)) - 1) >> shift)) ++bits;
    const int nbins = 1 << (3 * bits);
    h->counts.ensure((size_t)(nbins + 1) * sizeof(int));
    h->fill.ensure((size_t)(nbins + 1 + kCellPad) * sizeof(int));  // reused as tile_start (k_scan_apply pads its output)
    HIP_TRY(hipMemsetAsync(h->counts.p, 0, (size_t)(nbins + 1) * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_tile_count, dim3(pick_blocks(n, 256, 2048)), dim3(256), 0, h->stream, dc->pts(), ni, g, shift, h->counts.as<int>());
    // where every tile's queries start and where its batches start: one scan of packed {count, ceil(count / 32)} pairs
    const int ntiles = (nbins + kScanTile - 1) / kScanTile;
    h->tile_sums.ensure((size_t)ntiles * sizeof(unsigned long long));
    h->tmp.ensure(std::max(n * sizeof(float4), (size_t)(nbins + 1 + kCellPad) * sizeof(int)));  // (k_cell_rank is done with it) batches before every tile
    dc->n_batches_dev.ensure(sizeof(int));
    hipLaunchKernelGGL(k_scan2_tiles, dim3(ntiles), dim3(kScanBlock), 0, h->stream, h->counts.as<int>(), nbins, h->tile_sums.as<unsigned long long>());
    hipLaunchKernelGGL(k_scan2_tile_sums, dim3(1), dim3(kScanBlock), 0, h->stream, h->tile_sums.as<unsigned long long>(), ntiles);
    hipLaunchKernelGGL(k_scan2_apply, dim3(ntiles), dim3(kScanBlock), 0, h->stream, h->counts.as<int>(), nbins, h->tile_sums.as<unsigned long long>(), h->fill.as<int>(),
                       h->tmp.as<int>(), dc->n_batches_dev.as<int>());
    dc->qpts.ensure(n * sizeof(float4));
    hipLaunchKernelGGL(k_tile_place, dim3(pick_blocks(n, 256, 4096)), dim3(256), 0, h->stream, dc->pts(), ni, g, shift, dc->cells(),
                       h->fill.as<int>(), dc->qpts.as<float4>());
    // tile-aligned query batches
    dc->batches.ensure((n + 1) * sizeof(int2));
    hipLaunchKernelGGL(k_batch_fill, dim3((nbins + 255) / 256), dim3(256), 0, h->stream, h->counts.as<int>(), h->fill.as<int>(), h->tmp.as<int>(), nbins,
                       dc->batches.as<int2>());
    dc->batch_boxes.ensure((n + 1) * 6 * sizeof(float));
    hipLaunchKernelGGL(k_batch_boxes, dim3((unsigned)std::min<size_t>((n + 3) / 4, 4096)), dim3(256), 0, h->stream, dc->qpts.as<float4>(), dc->batches.as<int2>(), dc->n_batches_dev.as<int>(),
                       dc->batch_boxes.as<float>());  // (a wave per batch, grid-strided)
    HIP_TRY(hipMemcpyAsync(&dc->n_batches, dc->n_batches_dev.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  }
  // with a memoised voxel the occupancy read-back rides on the build's final synchronisation (it steers the memo for the
  // next cloud of this size class, not this build)
  if (auto_h && memo_hit) HIP_TRY(hipMemcpyAsync(&occ, h->occ.p, sizeof(occ), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipEventRecord(h->ev_b, h->stream));
  HIP_TRY(hipEventSynchronize(h->ev_b));
  if (auto_h) {  // remember the voxel for the next cloud of this size class (two entries: scan-sized and submap-sized)
    double next_h = (double)dc->grid.h;
    if (memo_hit) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      const double lam = (double)occ / (double)n, ratio = h->target_occupancy / std::max(lam, 1.0);
      if (!(ratio > 0.75 && ratio < 1.33)) next_h = std::max(0.01, next_h * std::min(4.0, std::max(0.25, std::pow(ratio, 1.0 / 1.5))));
    }
    int slot = -1;
    for (int i = 0; i < 4; ++i)  // the entry of this size class, else an empty one, else the oldest
      if (h->voxel_memo[i].second > 0.0 && (double)n > 0.5 * (double)h->voxel_memo[i].first && (double)n < 2.0 * (double)h->voxel_memo[i].first) slot = i;
    if (slot < 0)
      for (int i = 0; i < 4 && slot < 0; ++i)
        if (!(h->voxel_memo[i].second > 0.0)) slot = i;
    if (slot < 0) {
      slot = h->voxel_memo_next;
      h->voxel_memo_next = (h->voxel_memo_next + 1) % 4;
    }
    h->voxel_memo[slot] = {n, next_h};
  }
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
  dc->build_ms = ms;
  h->stats.index_build_ms = ms;
  HIP_TRY(hipGetLastError());
  return dc;
}

std::shared_ptr<DeviceCloud> upload_and_index(ngicp* h, const float* xyz, size_t n, size_t stride) {
  float mn[3], mx[3];
  stage_host_cloud(h, xyz, n, stride, mn, mx);
  return index_unsorted(h, n, mn, mx);
}

void ensure_inv_perm(ngicp* h, DeviceCloud& dc) {
  if (dc.has_inv) return;
  dc.inv_perm.ensure(dc.n * sizeof(int));
  hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)((dc.n + 255) / 256)), dim3(256), 0, h->stream, dc.perm.as<int>(), (int)dc.n, dc.inv_perm.as<int>());
  dc.has_inv = true;
}

void ensure_slot_ready(ngicp* h, Slot& s, const char* what) {
  if (!s.present) throw ArgError{NGICP_ERR_STATE, std::string("no ") + what + " cloud set"};
  if (s.dev) return;
  if (!s.host) throw ArgError{NGICP_ERR_STATE, std::string(what) + " cloud has no data"};
  s.dev = upload_and_index(h, s.host, s.n, s.stride);
}

// ------------------------------------------------------------------------------------------
// Covariances
// ------------------------------------------------------------------------------------------
template <int K>
void launch_cov(ngicp* h, DeviceCloud& dc, int k, int reg, double* out, size_t lo, size_t hi, hipStream_t s) {
  if (hi <= lo) return;
  const dim3 grid((unsigned)((hi - lo + kKnnPairs - 1) / kKnnPairs)), block(kKnnBlock);
  // window size by cloud size (see knn_take_window): a scan that does not fill the chip is as slow as one wave's chain of round
  // trips - wider windows; a large cloud is bound by what its waves fetch and insert - narrow ones
  if (dc.n < 160000)
    hipLaunchKernelGGL((k_covariances<K, 6>), grid, block, 0, s, dc.pts(), dc.cells(), dc.grid, (int)lo, (int)hi, k, reg, out);
  else
    hipLaunchKernelGGL((k_covariances<K, 4>), grid, block, 0, s, dc.pts(), dc.cells(), dc.grid, (int)lo, (int)hi, k, reg, out);
}

// covariances of the points at sorted positions [lo, hi) into `out` ([n][6], sorted order)
void launch_cov_range(ngicp* h, DeviceCloud& dc, double* out, size_t lo, size_t hi, hipStream_t s) {
  const int k = h->p.k, reg = h->p.regularization;
  if (k <= 0) throw ArgError{NGICP_ERR_ARG, "k must be positive"};
  if (k > 32 || (size_t)k > dc.n) throw ArgError{NGICP_ERR_K_TOO_LARGE, "k exceeds the cloud size or the engine limit of 32"};
  if (k <= 10)
    launch_cov<10>(h, dc, k, reg, out, lo, hi, s);
  else if (k <= 20)
    launch_cov<20>(h, dc, k, reg, out, lo, hi, s);
  else
    launch_cov<32>(h, dc, k, reg, out, lo, hi, s);
}

void compute_covs(ngicp* h, Slot& slot, CovSet& cs, const char* what) {
  ensure_slot_ready(h, slot, what);
  DeviceCloud& dc = *slot.dev;
  const int k = h->p.k;
  if (k <= 0) throw ArgError{NGICP_ERR_ARG, "k must be positive"};
  if (k > 32 || (size_t)k > dc.n) throw ArgError{NGICP_ERR_K_TOO_LARGE, "k exceeds the cloud size or the engine limit of 32"};
  auto buf = acquire_buf(h, h->device, dc.n * 6 * sizeof(double));
  HIP_TRY(hipEventRecord(h->ev_cov_a, h->stream));
  launch_cov_range(h, dc, buf->as<double>(), 0, dc.n, h->stream);
  HIP_TRY(hipEventRecord(h->ev_cov_b, h->stream));
  HIP_TRY(hipGetLastError());
  h->cov_timing_pending = true;  // no synchronisation here: the covariances are consumed on this same stream
  cs.data = buf;
  cs.n = dc.n;
  cs.order = slot.dev;
}

// make `cs` usable with cloud `dc` (same n): returns device pointer to [n][6] in dc's sorted order
const double* covs_for(ngicp* h, CovSet& cs, const std::shared_ptr<DeviceCloud>& dc) {
  if (cs.order.get() == dc.get()) return cs.data->as<double>();
  // covariances are logically indexed by ORIGINAL point index (the reference's vector index):
  // re-order from the donor cloud's sorted order to this cloud's sorted order
  ensure_inv_perm(h, *cs.order);
  auto buf = acquire_buf(h, h->device, dc->n * 6 * sizeof(double));
  hipLaunchKernelGGL(k_covs_reorder, dim3((unsigned)((dc->n + 255) / 256)), dim3(256), 0, h->stream, cs.data->as<double>(), cs.order->inv_perm.as<int>(), dc->perm.as<int>(),
                     (int)dc->n, buf->as<double>());
  cs.data = buf;
  cs.order = dc;
  return cs.data->as<double>();
}

void get_covs(ngicp* h, CovSet& cs, double* out) {
  if (cs.n == 0) return;
  h->scratch16.ensure(cs.n * 16 * sizeof(double));
  hipLaunchKernelGGL(k_covs_expand, dim3((unsigned)((cs.n + 255) / 256)), dim3(256), 0, h->stream, cs.data->as<double>(), cs.order->perm.as<int>(), (int)cs.n,
                     h->scratch16.as<double>());
  HIP_TRY(hipMemcpyAsync(out, h->scratch16.p, cs.n * 16 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
}

void set_covs(ngicp* h, Slot& slot, CovSet& cs, const double* in, size_t n, const char* what) {
  if (n == 0) {
    cs.clear();
    return;
  }
  // The reference accepts any vector; sizes are reconciled at align() (impl/nano_gicp_impl.hpp:163-168).
  // The packed image needs an ordering cloud: require the slot's cloud with the same size.
  if (!slot.present || slot.n != n) throw ArgError{NGICP_ERR_STATE, std::string("set covariances: ") + what + " cloud missing or of different size"};
  ensure_slot_ready(h, slot, what);
  h->scratch16.ensure(n * 16 * sizeof(double));
  HIP_TRY(hipMemcpyAsync(h->scratch16.p, in, n * 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  auto buf = acquire_buf(h, h->device, n * 6 * sizeof(double));
  hipLaunchKernelGGL(k_covs_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->scratch16.as<double>(), slot.dev->perm.as<int>(), (int)n, buf->as<double>());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
  cs.data = buf;
  cs.n = n;
  cs.order = slot.dev;
}

// ------------------------------------------------------------------------------------------
// Registration loop
// ------------------------------------------------------------------------------------------
// start / stop: events attached to the dispatch itself (they take the kernel's own begin / end timestamps: no extra packets in
// the stream, unlike hipEventRecord before and after), or null
std::mutex& persist_mutex(int device) {  // one persistent alignment per device at a time (its grid fills the device and its blocks wait for each other)
  static std::mutex m[64];
  return m[(unsigned)device % 64u];
}

int pass_impl() {
  static const int impl = std::getenv("NGICP_PASS_IMPL") ? std::atoi(std::getenv("NGICP_PASS_IMPL")) : 0;
  return impl;
}

void launch_pass(ngicp* h, const PassArgs& a, int nblocks, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr) {
  // 32-query batches, 2 lanes per query.  Two builds of the kernel: 3 waves per SIMD (129 VGPRs), and 4 (128 VGPRs, two spilled
  // dwords, 4 blocks per CU) for grids of more than two rounds of blocks, where the launch is bound by how many blocks pass through
  // the chip rather than by its slowest block.
  static const int force = std::getenv("NGICP_PASS_WPS") ? std::atoi(std::getenv("NGICP_PASS_WPS")) : 0;  // (A/B timing only)
  const int impl = pass_impl();  // 0: walks in global memory (default), 1: the staged search of ngicp_pass_st.h (round 3 experiment: exact, slower - DESIGN.md §5)
  const bool four = force ? force == 4 : nblocks > 2 * h->pass_slots;
  if (impl == 1) {
    PassArgs b = a;
    b.fused = 0;
    // cells that cover the distance gate around a query's own cell (its reach box is clamped there; beyond it the shell walk takes over)
    int need = kStGrowMax;
    if (h->p.max_corr_dist < 1e30) need = (int)std::ceil(h->p.max_corr_dist / (double)a.grid.h);
    b.stage_grow = std::max(1, std::min(kStGrowMax, need));
    if (force ? force == 4 : true)  // (128 VGPRs either way; the 4-wave build's smaller tables leave room for a fourth block per CU)
      hipExtLaunchKernelGGL((k_gicp_pass_st<4>), dim3(nblocks), dim3(256), 0, s, start, stop, 0, b);
    else
      hipExtLaunchKernelGGL((k_gicp_pass_st<3>), dim3(nblocks), dim3(256), 0, s, start, stop, 0, b);
    return;
  }
  // NGICP_QUEUE=1 (experiment, round 3): a grid of resident blocks that draw their groups from a counter (k_gicp_queue)
  static const int queue_env = std::getenv("NGICP_QUEUE") ? std::atoi(std::getenv("NGICP_QUEUE")) : 0;
  if (queue_env && !a.fused && !(a.mode & 4) && !a.dbg_stamps && !a.dbg_span && !a.dbg_qstats) {
    if (four)
      hipExtLaunchKernelGGL((k_gicp_queue<2, 4>), dim3((unsigned)std::min(nblocks, h->queue_slots[1])), dim3(256), 0, s, start, stop, 0, a);
    else
      hipExtLaunchKernelGGL((k_gicp_queue<2, 3>), dim3((unsigned)std::min(nblocks, h->queue_slots[0])), dim3(256), 0, s, start, stop, 0, a);
    return;
  }
  if (a.fused) {  // (the solver in the tail of the launch: a build of its own)
    if (four)
      hipExtLaunchKernelGGL((k_gicp_pass<2, 4, true>), dim3(nblocks), dim3(256), 0, s, start, stop, 0, a);
    else
      hipExtLaunchKernelGGL((k_gicp_pass<2, 3, true>), dim3(nblocks), dim3(256), 0, s, start, stop, 0, a);
    return;
  }
  if (four)
    hipExtLaunchKernelGGL((k_gicp_pass<2, 4>), dim3(nblocks), dim3(256), 0, s, start, stop, 0, a);
  else
    hipExtLaunchKernelGGL((k_gicp_pass<2, 3>), dim3(nblocks), dim3(256), 0, s, start, stop, 0, a);
}

struct LoopCtx {
  PassArgs pa;
  SolveArgs sa;
  int nblocks;
};

void prepare_loop(ngicp* h, LoopCtx& c) {
  ensure_slot_ready(h, h->src, "source");
  ensure_slot_ready(h, h->tgt, "target");
  // lazy covariances (impl/nano_gicp_impl.hpp:163-168)
  if (h->src_covs.n != h->src.dev->n) compute_covs(h, h->src, h->src_covs, "source");
  if (h->tgt_covs.n != h->tgt.dev->n) compute_covs(h, h->tgt, h->tgt_covs, "target");
  DeviceCloud& S = *h->src.dev;
  DeviceCloud& T = *h->tgt.dev;
  const size_t n = S.n;
  for (int i = 0; i < 2; ++i) {
    h->tpt[i].ensure(n * sizeof(float4));
    h->mahal[i].ensure(n * 6 * sizeof(double));
  }
  const int nblocks = std::max(1, (S.n_batches + 3) / 4);  // one block per group of four batches
  h->partials.ensure((size_t)kNumSlots * nblocks * sizeof(double));
  h->grp_order.ensure((size_t)nblocks * sizeof(int));
  h->grp_order_alt.ensure((size_t)nblocks * sizeof(int));
  h->grp_cost.ensure((size_t)nblocks * sizeof(int));
  {
    // (the persistent kernel's ring of per-pass views continues behind the state: one 256-byte entry per possible pass)
    const long ring = (long)std::max(1, h->p.max_iter) * std::max(1, h->p.lm_max_iter) + 2;
    h->state.ensure(sizeof(LmState) + (ring <= kMaxPersistPasses ? (size_t)ring * kViewWords * sizeof(int) : 0));
  }
  const int max_rows = std::max(1, h->p.max_iter) * std::max(1, h->p.lm_max_iter) + 1;
  if (h->trace.ensure_grew((size_t)max_rows * kTraceCols * sizeof(double))) h->trace_rows_dev = 0;  // an unfetched trace went with the old buffer
  h->sums.ensure(kPartialStride * sizeof(double));

  PassArgs& a = c.pa;
  a.qpts = S.qpts.as<float4>();
  a.batches = S.batches.as<int2>();
  a.batch_boxes = S.batch_boxes.as<float>();
  int* const order_buf[2] = {h->grp_order.as<int>(), h->grp_order_alt.as<int>()};
  int* const ctl = h->order_flag.as<int>();  // {order flag 0, ticket, gen, order flag 1}
  int* const order_flag[2] = {ctl, ctl + 3};
  a.grp_order = order_buf[h->order_sel];
  a.grp_cost = h->grp_cost.as<int>();
  a.n_batches = S.n_batches;
  a.cov_src = covs_for(h, h->src_covs, h->src.dev);
  a.n_src = (int)n;
  a.tgt = T.pts();
  a.tgt3 = T.xyz3();
  a.tgtp = T.xyzp();
  a.tgt_cell_start = T.cells();
  a.tgt_cell_box = (h->cell_boxes && T.has_boxes) ? T.cell_box.as<unsigned int>() + kCellPad : nullptr;
  a.cov_tgt = covs_for(h, h->tgt_covs, h->tgt.dev);
  a.grid = T.grid;
  for (int i = 0; i < 2; ++i) {
    a.tpt[i] = h->tpt[i].as<float4>();
    a.mahal[i] = h->mahal[i].as<double>();
  }
  a.gate_sq = h->p.max_corr_dist * h->p.max_corr_dist;
  {
    float f = (float)a.gate_sq;  // may round down or overflow to inf
    if ((double)f < a.gate_sq) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    a.gate_sq_f = f;
  }
  h->batch_far.ensure((size_t)S.n_batches + 16);
  a.batch_far = h->batch_far.as<unsigned char>();
  a.st = h->state.as<LmState>();
  a.partials = h->partials.as<double>();
  a.mode = 3;
  a.dbg_stamps = nullptr;
  a.dbg_qstats = nullptr;
  a.dbg_span = nullptr;
  a.order_valid = order_flag[h->order_sel];
  a.t_first = nullptr;
  a.fused = 0;
  a.persist = 0;
  a.first_pass = 0;
  a.max_passes = 0;
  a.ticket = ctl + 1;
  h->gen_lines.ensure((size_t)kGenLines * kGenStride * sizeof(int));
  a.gen = h->gen_lines.as<int>();
  {
    // rings worth staging: enough to cover the distance gate (the search never looks farther), at most kStageMaxGrow
    int need = kStageMaxGrow;
    if (h->p.max_corr_dist < 1e30) need = (int)std::ceil(h->p.max_corr_dist / (double)T.grid.h);
    a.stage_grow = h->stage_grow <= 0 ? 0 : std::max(1, std::min(std::min(kStageMaxGrow, h->stage_grow), need));  // 0: search straight from global memory
  }

  SolveArgs& s = c.sa;
  s.st = a.st;
  s.cfg.max_iterations = h->p.max_iter;
  s.cfg.lm_max_iterations = h->p.lm_max_iter;
  s.cfg.optimizer = h->p.optimizer;
  s.cfg.rot_eps = h->p.rot_eps;
  s.cfg.trans_eps = h->p.trans_eps;
  s.cfg.lm_init_lambda_factor = h->p.lm_init_lambda_factor;
  s.partials = a.partials;
  s.nblocks = nblocks;
  s.grp_order = order_buf[h->order_sel];
  s.grp_cost = h->grp_cost.as<int>();
  s.trace = h->trace.as<double>();
  s.max_trace_rows = max_rows;
  s.mode = 0;
  s.sums_out = nullptr;
  s.dbg_stamps = nullptr;
  s.progress_host = nullptr;
  s.final_host = nullptr;
  s.order_valid = order_flag[h->order_sel];
  s.t_first = nullptr;
  s.persist = 0;
  s.pass_ticks = nullptr;
  s.st_out = nullptr;
  s.nrows = 0;
  a.crow_in = nullptr;
  a.crow_out = nullptr;
  a.cluster_ticket = nullptr;
  a.done_flag = nullptr;
  c.nblocks = nblocks;
  h->stats.lanes_per_query = 2;
  h->stats.voxel_size = T.grid.h;
  h->stats.grid_dims[0] = T.grid.nx;
  h->stats.grid_dims[1] = T.grid.ny;
  h->stats.grid_dims[2] = T.grid.nz;
}

void init_state_from_pose(LmState& st, const Pose& x0) {
  std::memset(&st, 0, sizeof(st));
  st.hot.x0 = x0;
  st.hot.xi = x0;
  pose_identity(st.hot.delta);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) st.xi_f[r * 4 + c] = (float)x0.R[r * 3 + c];
    st.xi_f[r * 4 + 3] = (float)x0.t[r];
  }
  std::memcpy(&st.view[kViewXi], &st.hot.xi, sizeof(Pose));
  std::memcpy(&st.view[kViewXiF], st.xi_f, sizeof(st.xi_f));
  static_assert(sizeof(Pose) == 24 * sizeof(int) && kViewXiF == 24 && sizeof(LmState::xi_f) == 12 * sizeof(int), "LmState::view layout");
  st.hot.lambda = -1.0;  // impl/lsq_registration_impl.hpp:92
  st.hot.nu = 2.0;
  for (int i = 0; i < 6; ++i) st.hot.final_H[i * 6 + i] = 1.0;
}

Pose pose_from_colmajor_f(const float m[16]) {  // Isometry3d(guess.cast<double>()), impl/lsq_registration_impl.hpp:90
  Pose p;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) p.R[r * 3 + c] = (double)m[c * 4 + r];
    p.t[r] = (double)m[12 + r];
  }
  return p;
}
Pose pose_from_colmajor_d(const double m[16]) {
  Pose p;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) p.R[r * 3 + c] = m[c * 4 + r];
    p.t[r] = m[12 + r];
  }
  return p;
}
void pose_to_colmajor_f(const Pose& p, float m[16]) {  // x0.cast<float>().matrix(), :113
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) m[c * 4 + r] = (float)p.R[r * 3 + c];
    m[12 + r] = (float)p.t[r];
    m[r * 4 + 3] = 0.f;
  }
  m[15] = 1.f;
}

// h->out_xyz (packed xyz on the device) -> host xyz at a byte stride
void download_xyz(ngicp* h, size_t n, float* out, size_t out_stride) {
  if (out_stride == 12) {
    HIP_TRY(hipMemcpyAsync(out, h->out_xyz.p, n * 12, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } else {
    std::vector<float> tmp(n * 3);
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->out_xyz.p, n * 12, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < n; ++i) {
      float* o = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(out) + i * out_stride);
      o[0] = tmp[i * 3 + 0];
      o[1] = tmp[i * 3 + 1];
      o[2] = tmp[i * 3 + 2];
    }
  }
  HIP_TRY(hipGetLastError());
}

// device-resident cloud, transformed by a float matrix (pcl::transformPointCloud), to the host in ORIGINAL point order
void download_transformed(ngicp* h, DeviceCloud& dc, const float T_colmajor[16], float* out, size_t out_stride) {
  const size_t n = dc.n;
  h->tfinal.ensure(16 * sizeof(float));
  h->out_xyz.ensure(n * 3 * sizeof(float));
  HIP_TRY(hipMemcpyAsync(h->tfinal.p, T_colmajor, 16 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_transform_sorted_to_original, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, dc.pts(), (int)n, h->tfinal.as<float>(), h->out_xyz.as<float>());
  download_xyz(h, n, out, out_stride);
}

void do_align(ngicp* h, const float guess[16], float* aligned, size_t out_stride) {
  const double t_begin = now_ms();
  h->hook_valid = 0;
  h->converged = 0;
  h->nr_iterations = 0;
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(h->final_T, I, sizeof(I));  // PCL align(): final_transformation_ = Identity before computeTransformation
  LoopCtx c;
  prepare_loop(h, c);
  LmState st;
  init_state_from_pose(st, pose_from_colmajor_f(guess));
  // the launch order the previous align ended with is still a good guess when the source index is the same one
  // (same batches; the costs come mostly from where the batches lie): the first pass then starts sorted as well
  // (the flag lives in a device word of its own: the solver sets it when an order is complete, the host only clears it when the
  // source index or the group count changed - it never has to read it back)
  if (!(h->order_src == h->src.dev.get() && h->order_groups == c.sa.nblocks)) HIP_TRY(hipMemsetAsync(h->order_flag.p, 0, 4 * sizeof(int), h->stream));  // (both flags)
  // NGICP_ORDER=xcd (experiment): instead of the cost-sorted launch order, a FIXED order that hands every XCD (blocks b, b + 8, ...
  // are observed to share one) a contiguous eighth of the Morton-ordered groups: each XCD's L2 then sees an eighth of the target.
  static const bool xcd_order = std::getenv("NGICP_ORDER") && std::string(std::getenv("NGICP_ORDER")) == "xcd";
  if (xcd_order) {
    const int nb = c.nblocks, per = (nb + 7) / 8;
    std::vector<int> ord((size_t)nb);
    std::vector<int> lists[8];
    for (int g = 0; g < nb; ++g) lists[std::min(7, g / per)].push_back(g);
    size_t taken[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b = 0;
    for (int placed = 0; placed < nb; ++b) {  // block b belongs to XCD b % 8: the next group of that XCD's list, or of the fullest one left
      int x = b % 8;
      if (taken[x] >= lists[x].size()) {
        x = 0;
        for (int y = 1; y < 8; ++y)
          if (lists[y].size() - taken[y] > lists[x].size() - taken[x]) x = y;
      }
      ord[(size_t)placed++] = lists[x][taken[x]++];
    }
    HIP_TRY(hipMemcpyAsync(const_cast<int*>(c.pa.grp_order), ord.data(), (size_t)nb * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int one = 1;
    HIP_TRY(hipMemcpy(const_cast<int*>(c.pa.order_valid), &one, sizeof(int), hipMemcpyHostToDevice));
    c.sa.grp_order = nullptr;  // the solver leaves the order alone
  }
  c.pa.mode = (h->p.optimizer == NGICP_OPT_GAUSS_NEWTON) ? 2 : 3;
  // Whether the FIRST pass lists the region rows of every batch (later passes list for the batches that looked beyond ring 1 in the
  // pass before): it pays where many queries do (100k -> 500k with DLO's settings: 22 % of the batches, scan-to-submap 0.67 -> 0.64 ms)
  // and costs where few do (250k -> 2M: first pass 107 -> 72 us without).  Decided from the share of queries the previous alignment
  // of this handle served through lists; yes when there was none.
  if (h->prev_staged_fraction < 0.0 || h->prev_staged_fraction >= 0.12) c.pa.mode |= 32;
  if (const char* dbg = std::getenv("NGICP_DEBUG_MODE")) c.pa.mode |= (std::atoi(dbg) & (8 | 16));  // timing experiments only
  if (h->p.max_iter <= 0) st.hot.done = 1;
  h->pin_state[0] = st;
  HIP_TRY(hipMemcpyAsync(h->state.p, &h->pin_state[0], sizeof(st), hipMemcpyHostToDevice, h->stream));

  const char* stamp_path = std::getenv("NGICP_DEBUG_STAMPS");  // diagnostic only
  if (stamp_path) {
    h->dbg.ensure((size_t)(c.nblocks + 1) * 4 * kStampStride * sizeof(unsigned long long));  // (k_gicp_head has one block more)
    HIP_TRY(hipMemsetAsync(h->dbg.p, 0, (size_t)(c.nblocks + 1) * 4 * kStampStride * sizeof(unsigned long long), h->stream));
    c.pa.dbg_stamps = h->dbg.as<unsigned long long>();
  }
  if (std::getenv("NGICP_DEBUG_SOLVE")) {  // diagnostic only: s_memtime stamps of the last solver launch, printed after the align
    h->dbg_s.ensure(8 * sizeof(unsigned long long));
    HIP_TRY(hipMemsetAsync(h->dbg_s.p, 0, 8 * sizeof(unsigned long long), h->stream));
    c.sa.dbg_stamps = h->dbg_s.as<unsigned long long>();
  }
  const char* span_path = std::getenv("NGICP_DEBUG_SPAN");  // diagnostic only: when and where every block of the last pass ran
  if (span_path) {
    h->dbg_span.ensure((size_t)c.nblocks * 4 * sizeof(unsigned long long));
    HIP_TRY(hipMemsetAsync(h->dbg_span.p, 0, (size_t)c.nblocks * 4 * sizeof(unsigned long long), h->stream));
    c.pa.dbg_span = h->dbg_span.as<unsigned long long>();
  }
  const char* qstat_path = std::getenv("NGICP_DEBUG_QSTATS");  // diagnostic only: per-query search statistics of the last pass
  if (qstat_path) {
    h->dbg_q.ensure((size_t)c.pa.n_src * 2 * sizeof(int4));
    HIP_TRY(hipMemsetAsync(h->dbg_q.p, 0, (size_t)c.pa.n_src * 2 * sizeof(int4), h->stream));
    c.pa.dbg_qstats = h->dbg_q.as<int4>();
  }
  const long max_passes = (h->p.optimizer == NGICP_OPT_GAUSS_NEWTON) ? (long)h->p.max_iter : (long)h->p.max_iter * std::max(1, h->p.lm_max_iter) + 1;
  // The host feeds (pass, solve) pairs to the stream and never blocks on it inside the loop: the solver publishes its progress
  // {passes done, done flag} in PINNED host memory (one system-scope store), the host keeps `depth` pairs in flight and stops
  // feeding when it sees the flag.  At most `depth` pairs are enqueued in vain (they return at once: the state says done);
  // round 1 polled a copied flag one chunk of four pairs behind and wasted up to eight.
  const int depth = h->chunk_pairs;
  *h->h_progress = 0;
  c.sa.progress_host = h->h_progress;
  c.sa.final_host = h->pin_final;
  c.sa.t_first = h->t_first.as<unsigned long long>();
  c.pa.t_first = h->t_first.as<unsigned long long>();
  // NGICP_FUSED=1: one dispatch per iteration - the last block of the pass reduces and advances the optimiser (PassArgs::fused).
  // Measured on MI355X (round 3, profiles/r03_fused_solver.txt): bit-identical results, but no faster than the separate launch (c3
  // 46.3 vs 45.4 us per iteration, c5 70 vs 56): the write-through rows come back from memory, not from L2, through ONE CU
  // (8.7 k cycles for 222 KB against 5.5 k in k_lm_solve), plus the acquire (~1.7 us) - so the default stays two launches.
  static const bool fused_env = std::getenv("NGICP_FUSED") && std::atoi(std::getenv("NGICP_FUSED")) != 0;
  if (fused_env && pass_impl() == 0) {
    c.pa.fused = 1;
    c.pa.sa = c.sa;
    HIP_TRY(hipMemsetAsync(c.pa.ticket, 0, sizeof(int), h->stream));
  }
  long launched = 0;
  bool finished = (h->p.max_iter <= 0), persist_done = false;
  float persist_loop_ms = 0.f;
  const double t_loop = now_ms();
  unsigned long spins = 0;
  // ---- NGICP_PERSIST=1 (experiment, round 3): ONE launch for the whole alignment (k_gicp_persist): as many blocks as are resident
  //      together, each keeping its groups pass after pass; the last block to finish a pass steps the optimiser and releases the next
  //      one.  The idea: no second dispatch, no kernel boundaries - and with them no cold caches (a launch boundary drops every L2, and
  //      ~9 of 10 L2 read requests of a pass go out to the fabric: profiles/r03_c3_pass_counters.json).  Measured (profiles/
  //      r03_persistent_kernel.txt): bit-identical results; the groups run 4 % faster, but every hop of the grid-wide meeting (rows
  //      written through, ticket, state, release word, view) is a ~1-2 us round trip to memory, as long as the launches they replace:
  //      c3 50.7 us per iteration against 48.2, c5 76 against 58 (it has no 4-waves build).  So the default stays one launch per pass.
  //      Launched cooperatively: the runtime guarantees that the grid is resident as a whole (the blocks wait for each other), and
  //      one alignment per device at a time takes this route. ----
  bool persist_lock = false;
  const bool want_persist = h->persist && !finished && h->persist_slots > 0 && pass_impl() == 0 && !c.pa.fused && !xcd_order && !stamp_path && !span_path && !qstat_path &&
                            !c.sa.dbg_stamps && max_passes + 1 <= kMaxPersistPasses;
  if (want_persist) persist_lock = persist_mutex(h->device).try_lock();
  if (persist_lock) {
    struct Unlock {
      std::mutex& m;
      ~Unlock() { m.unlock(); }
    } unlock{persist_mutex(h->device)};
    int* const ctl = h->order_flag.as<int>();
    const int sel = h->order_sel;
    // {flag 0, ticket, gen, flag 1}: ticket and gen start at zero, and so does the flag of the buffer this alignment's solver will fill
    HIP_TRY(hipMemsetAsync(ctl + (sel == 0 ? 1 : 0), 0, 3 * sizeof(int), h->stream));
    HIP_TRY(hipMemsetAsync(h->gen_lines.p, 0, (size_t)kGenLines * kGenStride * sizeof(int), h->stream));
    PassArgs pa = c.pa;
    pa.fused = 1;
    pa.persist = 1;
    pa.max_passes = (int)max_passes;
    pa.sa = c.sa;
    pa.sa.persist = 1;
    pa.sa.grp_order = sel == 0 ? h->grp_order_alt.as<int>() : h->grp_order.as<int>();
    pa.sa.order_valid = sel == 0 ? ctl + 3 : ctl;
    pa.sa.pass_ticks = (h->profiling && max_passes <= kMaxTickPasses) ? h->pin_ticks : nullptr;
    const int grid = std::min(c.nblocks, h->persist_slots);
    void* kargs[] = {&pa};
    static const bool coop = !(std::getenv("NGICP_PERSIST_COOP") && std::atoi(std::getenv("NGICP_PERSIST_COOP")) == 0);
    bool launched_ok = true;
    if (coop) {
      const hipError_t le = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_gicp_persist<2, 3>), dim3((unsigned)grid), dim3(256), kargs, 0, h->stream);
      if (le != hipSuccess) {  // (e.g. the runtime finds the grid too large to be resident: one launch per pass then)
        (void)hipGetLastError();
        launched_ok = false;
      }
    } else {  // (A/B timing only: an ordinary launch relies on nothing else running on the device)
      hipLaunchKernelGGL((k_gicp_persist<2, 3>), dim3((unsigned)grid), dim3(256), 0, h->stream, pa);
    }
    bool ok = false;
    while (launched_ok) {
      const int prog = __atomic_load_n(h->h_progress, __ATOMIC_ACQUIRE);
      if (prog & kProgressDone) {
        ok = true;
        break;
      }
      if ((++spins & 0x3fff) == 0) {
        if (hipStreamQuery(h->stream) == hipSuccess) {  // the kernel has left: done flag (then it is in memory by now), or its blocks gave up waiting
          ok = (__atomic_load_n(h->h_progress, __ATOMIC_ACQUIRE) & kProgressDone) != 0;
          break;
        }
        if (now_ms() - t_loop > 30000.0) throw ArgError{NGICP_ERR_HIP, "the registration loop did not finish within 30 s"};
      }
      if (h->host_wait) sched_yield(); else __builtin_ia32_pause();
    }
    if (ok) {
      finished = true;
      st.hot = *h->pin_final;
      persist_loop_ms = (float)((double)(st.hot.t_done - st.hot.t_first) * 1e-5);
      persist_done = true;
      if (st.hot.passes >= 3 && c.nblocks <= kMaxOrderGroups) h->order_sel = sel ^ 1;  // the solver of the third pass left a fresh order in the other buffer
    } else {
      // (never seen: the blocks' bounded wait ran out - e.g. the grid was not resident as a whole.  The state goes back to the guess and the
      // alignment takes one launch per pass.)
      std::fprintf(stderr, launched_ok ? "ngicp: the persistent registration kernel gave up waiting; falling back to one launch per pass\n"
                                       : "ngicp: the persistent registration kernel could not be launched; falling back to one launch per pass\n");
      h->persist = 0;
      HIP_TRY(hipMemsetAsync(h->order_flag.p, 0, 4 * sizeof(int), h->stream));
      HIP_TRY(hipMemcpyAsync(h->state.p, &h->pin_state[0], sizeof(st), hipMemcpyHostToDevice, h->stream));
      *h->h_progress = 0;
    }
  }
  // ---- NGICP_HEAD=1: no solver launch at all (k_gicp_head): one launch per iteration, and one more whose head consumes the last pass ----
  const bool head_mode = h->head && !finished && pass_impl() == 0 && !c.pa.fused && !xcd_order && !span_path && !qstat_path && !c.sa.dbg_stamps &&
                         c.nblocks <= 2 * h->pass_slots;  // (grids of more rounds: the head's few microseconds are paid once per round)
  LmState* head_state[2] = {nullptr, nullptr};
  double* head_crow[2] = {nullptr, nullptr};
  int *head_tickets = nullptr, *head_done = nullptr, *head_order[2] = {nullptr, nullptr}, *head_flag[2] = {nullptr, nullptr};
  if (head_mode) {
    h->state_alt.ensure(sizeof(LmState));
    const size_t crow_bytes = (size_t)kSolveRowSubsets * kNumSlots * sizeof(double);
    h->head_ws.ensure(64 + 128 + 2 * crow_bytes);
    HIP_TRY(hipMemsetAsync(h->head_ws.p, 0, 64 + 128 + 2 * crow_bytes, h->stream));
    unsigned char* ws = h->head_ws.as<unsigned char>();
    head_done = reinterpret_cast<int*>(ws);
    head_tickets = reinterpret_cast<int*>(ws + 64);
    head_crow[0] = reinterpret_cast<double*>(ws + 192);
    head_crow[1] = reinterpret_cast<double*>(ws + 192 + crow_bytes);
    head_state[0] = h->state.as<LmState>();
    head_state[1] = h->state_alt.as<LmState>();
    int* const ctl = h->order_flag.as<int>();
    head_order[0] = h->grp_order.as<int>();
    head_order[1] = h->grp_order_alt.as<int>();
    head_flag[0] = ctl;
    head_flag[1] = ctl + 3;
  }
  const long max_launches = head_mode ? max_passes + 1 : max_passes;
  while (!finished && launched < max_launches) {
    const int prog = *reinterpret_cast<volatile int*>(h->h_progress);
    if (prog & kProgressDone) break;
    if (launched - (long)(prog & kProgressMask) >= depth) {  // enough in flight: wait for the device to catch up
      if ((++spins & (h->host_wait ? 0xfff : 0xfffff)) == 0 && now_ms() - t_loop > 30000.0) throw ArgError{NGICP_ERR_HIP, "the registration loop did not finish within 30 s"};
      if (h->host_wait) sched_yield(); else __builtin_ia32_pause();
      continue;
    }
    const bool timed = h->profiling && launched % h->prof_stride == h->prof_stride / 2 && (size_t)(2 * launched + 1) < h->prof_events.size();
    if (head_mode) {
      // k_gicp_head: launch i reads state / subset rows / launch order [i & 1] and leaves the next ones in [(i + 1) & 1]
      const int par = (int)(launched & 1);
      PassArgs pa = c.pa;
      pa.st = head_state[par];
      pa.crow_in = head_crow[par];
      pa.crow_out = head_crow[par ^ 1];
      pa.cluster_ticket = head_tickets;
      pa.done_flag = head_done;
      pa.grp_order = head_order[par];
      pa.order_valid = head_flag[par];
      pa.sa = c.sa;
      pa.sa.st = head_state[par];
      pa.sa.st_out = head_state[par ^ 1];
      pa.sa.partials = head_crow[par];
      pa.sa.nrows = kSolveRowSubsets;
      pa.sa.grp_order = head_order[par ^ 1];
      pa.sa.order_valid = head_flag[par ^ 1];
      hipExtLaunchKernelGGL((k_gicp_head<2, 3>), dim3((unsigned)c.nblocks + 1), dim3(256), 0, h->stream, timed ? h->prof_events[2 * launched] : nullptr,
                            timed ? h->prof_events[2 * launched + 1] : nullptr, 0, pa);
      ++launched;
      continue;
    }
    static const bool persist_one = std::getenv("NGICP_PERSIST_ONE") != nullptr;  // A/B only: the persistent kernel's code, one launch per pass
    if (persist_one && h->persist_slots > 0 && max_passes + 1 <= kMaxPersistPasses) {
      PassArgs pa = c.pa;
      pa.fused = 1;
      pa.persist = std::atoi(std::getenv("NGICP_PERSIST_ONE")) == 2 ? 2 : 1;
      pa.first_pass = (int)launched;
      pa.max_passes = 1;
      pa.sa = c.sa;
      pa.sa.persist = 1;
      HIP_TRY(hipMemsetAsync(pa.ticket, 0, sizeof(int), h->stream));
      HIP_TRY(hipMemsetAsync(h->gen_lines.p, 0, (size_t)kGenLines * kGenStride * sizeof(int), h->stream));
      hipExtLaunchKernelGGL((k_gicp_persist<2, 3>), dim3((unsigned)std::min(c.nblocks, h->persist_slots)), dim3(256), 0, h->stream,
                            timed ? h->prof_events[2 * launched] : nullptr, timed ? h->prof_events[2 * launched + 1] : nullptr, 0, pa);
      ++launched;
      continue;
    }
    launch_pass(h, c.pa, c.nblocks, h->stream, timed ? h->prof_events[2 * launched] : nullptr, timed ? h->prof_events[2 * launched + 1] : nullptr);
    if (!c.pa.fused) hipLaunchKernelGGL(k_lm_solve, dim3(1), dim3(kSolveThreads), 0, h->stream, c.sa);
    ++launched;
  }
  float loop_ms = 0.f;
  if (persist_done) {
    loop_ms = persist_loop_ms;
  } else if (finished) {
    HIP_TRY(hipStreamSynchronize(h->stream));  // (max_iterations <= 0: nothing was launched; the state is the initial one)
  } else {
    // The solver writes the final state image into pinned memory and THEN raises the done flag (system-scope release): no copy,
    // no event, no stream synchronisation - the few launches enqueued ahead return at once behind the host's back, and whatever
    // this handle enqueues next is ordered behind them on its stream.
    for (;;) {
      const int prog = __atomic_load_n(h->h_progress, __ATOMIC_ACQUIRE);
      if (prog & kProgressDone) break;
      if (launched >= max_launches && launched - (long)(prog & kProgressMask) <= 0) break;  // (cannot happen: the last possible pass sets done)
      if ((++spins & (h->host_wait ? 0xfff : 0xfffff)) == 0 && now_ms() - t_loop > 30000.0) throw ArgError{NGICP_ERR_HIP, "the registration loop did not finish within 30 s"};
      if (h->host_wait) sched_yield(); else __builtin_ia32_pause();
    }
    st.hot = *h->pin_final;
    loop_ms = (float)((double)(st.hot.t_done - st.hot.t_first) * 1e-5);  // 100 MHz ticks -> ms
    // (k_gicp_head: the head of launch `passes` ended the alignment and left the final state in the buffer it does not read; the handle's
    // other entry points look for it in h->state)
    if (head_mode && ((st.hot.passes + 1) & 1)) HIP_TRY(hipMemcpyAsync(h->state.p, h->state_alt.p, sizeof(LmState), hipMemcpyDeviceToDevice, h->stream));
  }
  HIP_TRY(hipGetLastError());

  if (stamp_path) {
    std::vector<unsigned long long> hs((size_t)(c.nblocks + 1) * 4 * kStampStride);
    HIP_TRY(hipMemcpy(hs.data(), h->dbg.p, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(stamp_path, "wb")) {
      std::fwrite(hs.data(), sizeof(unsigned long long), hs.size(), f);
      std::fclose(f);
    }
  }
  if (c.sa.dbg_stamps) {
    unsigned long long ts[8];
    HIP_TRY(hipMemcpy(ts, h->dbg_s.p, sizeof(ts), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "k_lm_solve stamps (cycles since entry): loads issued %llu, reduced %llu, state in registers %llu, lm_advance %llu, accept path %llu, stored %llu; launch-order section (wave 1) %llu cycles\n",
                 ts[1] - ts[0], ts[2] - ts[0], ts[3] - ts[0], ts[4] - ts[0], ts[5] - ts[0], ts[6] - ts[0], ts[7]);
  }
  if (span_path) {
    std::vector<unsigned long long> hs((size_t)c.nblocks * 4);
    HIP_TRY(hipMemcpy(hs.data(), h->dbg_span.p, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(span_path, "wb")) {
      std::fwrite(hs.data(), sizeof(unsigned long long), hs.size(), f);
      std::fclose(f);
    }
  }
  if (const char* cost_path = std::getenv("NGICP_DEBUG_COSTS")) {  // diagnostic only: the groups' durations in the last pass (cycles >> 4), the launch order, the partial rows
    std::vector<int> hc((size_t)c.nblocks * 2);
    HIP_TRY(hipMemcpy(hc.data(), h->grp_cost.p, (size_t)c.nblocks * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hc.data() + c.nblocks, h->grp_order.p, (size_t)c.nblocks * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<double> hp((size_t)c.nblocks * kNumSlots);
    HIP_TRY(hipMemcpy(hp.data(), h->partials.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(cost_path, "wb")) {
      std::fwrite(hc.data(), sizeof(int), hc.size(), f);
      std::fwrite(hp.data(), sizeof(double), hp.size(), f);
      std::fclose(f);
    }
  }
  if (qstat_path) {
    std::vector<int> hq((size_t)c.pa.n_src * 8);
    HIP_TRY(hipMemcpy(hq.data(), h->dbg_q.p, hq.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(qstat_path, "wb")) {
      std::fwrite(hq.data(), sizeof(int), hq.size(), f);
      std::fclose(f);
    }
  }
  pose_to_colmajor_f(st.hot.x0, h->final_T);
  h->converged = st.hot.converged;
  h->nr_iterations = st.hot.nr_iterations;
  for (int r = 0; r < 6; ++r)
    for (int cc = 0; cc < 6; ++cc) h->final_hessian[cc * 6 + r] = st.hot.final_H[r * 6 + cc];
  if (st.hot.lm_failed) std::fprintf(stderr, "lm not converged!!\n");  // impl/lsq_registration_impl.hpp:106
  h->trace_host.clear();  // fetched on demand (ngicp_get_lm_trace): a diagnostic should not cost every align a synchronous copy
  h->trace_rows_dev = (size_t)st.hot.n_trace;

  if (aligned) download_transformed(h, *h->src.dev, h->final_T, aligned, out_stride);  // K5: pcl::transformPointCloud(*input_, output, final_transformation_)
  ngicp_stats& s = h->stats;
  s.loop_ms = loop_ms;
  if (st.hot.have_lin) h->hook_valid = 2;  // ngicp_get_correspondences: the correspondences of the last adopted linearisation
  h->order_src = h->src.dev.get();  // (what the order flag on the device, if set, refers to)
  h->order_groups = c.sa.nblocks;
  s.passes = st.hot.passes;
  s.outer_iterations = st.hot.nr_iterations + 1;
  s.lm_trials = st.hot.n_trace;
  s.mean_candidates = st.hot.passes > 0 ? st.hot.cand_total / ((double)st.hot.passes * (double)h->src.dev->n) : 0.0;
  s.valid_fraction = st.hot.passes > 0 ? st.hot.valid_total / ((double)st.hot.passes * (double)h->src.dev->n) : 0.0;
  s.staged_fraction = st.hot.passes > 0 ? st.hot.staged_total / ((double)st.hot.passes * (double)h->src.dev->n) : 0.0;
  if (st.hot.passes > 1) h->prev_staged_fraction = s.staged_fraction;
  s.pass_ms_total = 0.0;
  if (h->profiling && persist_done) {
    // the persistent kernel's own stamps (100 MHz): a pass lasts from its release (the first: the alignment's first stamp) to the arrival
    // of its last block; the optimiser's step and the release that follows are not part of it
    const long timed = std::min<long>(st.hot.passes, kMaxTickPasses);
    int counted = 0;
    if ((long)h->p.max_iter * std::max(1, h->p.lm_max_iter) + 1 <= kMaxTickPasses) {
      for (long i = 0; i < timed; ++i) {
        const unsigned long long from = i == 0 ? st.hot.t_first : h->pin_ticks[2 * (i - 1) + 1], to = h->pin_ticks[2 * i];
        if (to > from) {
          s.pass_ms_total += (double)(to - from) * 1e-5;
          ++counted;
        }
      }
    }
    s.passes_timed = counted;
    if (std::getenv("NGICP_DEBUG_TICKS")) {  // diagnostic only: every pass and every step of the alignment, in microseconds
      std::fprintf(stderr, "persistent kernel, pass / step us:");
      for (long i = 0; i < timed; ++i) {
        const unsigned long long from = i == 0 ? st.hot.t_first : h->pin_ticks[2 * (i - 1) + 1];
        std::fprintf(stderr, " %.1f/%.1f", (double)(h->pin_ticks[2 * i] - from) * 1e-2, (double)(h->pin_ticks[2 * i + 1] - h->pin_ticks[2 * i]) * 1e-2);
      }
      std::fprintf(stderr, "\n");
    }
  } else if (h->profiling) {
    // HIP events on the handle's own stream around every pass launch that did work
    const long timed = std::min<long>(st.hot.passes, (long)h->prof_events.size() / 2);
    int counted = 0;
    for (long i = h->prof_stride / 2; i < timed; i += h->prof_stride) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->prof_events[2 * i], h->prof_events[2 * i + 1]) == hipSuccess) {
        s.pass_ms_total += ms;
        ++counted;
      }
    }
    s.passes_timed = counted;
  } else {
    s.passes_timed = 0;
  }
  s.n_src = (long long)h->src.dev->n;
  s.n_tgt = (long long)h->tgt.dev->n;
  s.host_wait_spins = (long long)spins;
  s.align_ms = now_ms() - t_begin;
}

template <class F>
int guarded(ngicp* h, F&& f) {
  if (!h) return NGICP_ERR_ARG;
  try {
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) throw HipError{e, "hipSetDevice", __FILE__, __LINE__};
    f();
    return NGICP_OK;
  } catch (const HipError& e) {
    char buf[512];
    std::snprintf(buf, sizeof(buf), "HIP error %d (%s) in `%s` at %s:%d", (int)e.code, hipGetErrorString(e.code), e.what, e.file, e.line);
    h->err = buf;
    (void)hipGetLastError();
    return NGICP_ERR_HIP;
  } catch (const ArgError& e) {
    h->err = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    h->err = e.what();
    return NGICP_ERR_ARG;
  } catch (...) {
    h->err = "unknown error";
    return NGICP_ERR_ARG;
  }
}

int set_cloud(ngicp* h, Slot& slot, const float* xyz, size_t n, size_t stride, uint64_t identity, bool build_now) {
  return guarded(h, [&] {
    if (!xyz && n) throw ArgError{NGICP_ERR_ARG, "null cloud pointer"};
    if (identity != 0 && slot.present && slot.identity == identity) return;  // pointer-identity early-out
    slot.clear();
    slot.present = true;
    slot.host = xyz;
    slot.n = n;
    slot.stride = stride;
    slot.identity = identity;
    if (build_now) {
      try {
        slot.dev = upload_and_index(h, xyz, n, stride);
      } catch (...) {
        slot.clear();  // a rejected cloud leaves the slot empty
        throw;
      }
    }
  });
}

}  // namespace

// =============================================================================================
extern "C" {

const char* ngicp_version(void) { return "ngicp-hip 0.1 (gfx950)"; }

int ngicp_create(int device, ngicp_t** out) {
  if (!out) return NGICP_ERR_ARG;
  *out = nullptr;
  try {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
      g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
      (void)hipGetLastError();
      return NGICP_ERR_HIP;
    }
    if (device < 0 || device >= count) {
      g_create_error = "device index out of range";
      return NGICP_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<ngicp> h(new ngicp);
    h->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&h->ev_a));
    HIP_TRY(hipEventCreate(&h->ev_b));
    HIP_TRY(hipEventCreate(&h->ev_cov_a));
    HIP_TRY(hipEventCreate(&h->ev_cov_b));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_fence, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_state), 2 * sizeof(LmState), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_final), sizeof(LmHot), hipHostMallocDefault));
    h->order_flag.ensure(64);
    h->t_first.ensure(64);
    HIP_TRY(hipMemset(h->order_flag.p, 0, 64));
    HIP_TRY(hipMemset(h->t_first.p, 0, 64));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_progress), sizeof(int), hipHostMallocDefault));
    *h->h_progress = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_shard_done), kShardSlots * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < kShardSlots; ++i) {
      h->h_shard_done[i] = 0;
      HIP_TRY(hipEventCreateWithFlags(&h->ev_shard[i], hipEventDisableTiming));
    }
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(h->final_T, I, sizeof(I));
    std::memset(h->final_hessian, 0, sizeof(h->final_hessian));
    for (int i = 0; i < 6; ++i) h->final_hessian[i * 6 + i] = 1.0;  // impl/lsq_registration_impl.hpp:62
    {
      int cus = 0, per_cu = 0;
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gicp_pass<2, 3>, 256, 0));
      h->pass_slots = std::max(1, cus) * std::max(1, per_cu);
      int per_cu_p = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_p, k_gicp_persist<2, 3>, 256, 0) == hipSuccess) h->persist_slots = std::max(0, cus) * std::max(0, per_cu_p);
      int q3 = 0, q4 = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q3, k_gicp_queue<2, 3>, 256, 0) == hipSuccess && q3 > 0) h->queue_slots[0] = cus * q3;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q4, k_gicp_queue<2, 4>, 256, 0) == hipSuccess && q4 > 0) h->queue_slots[1] = cus * q4;
      int coop = 0;
      if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess || !coop) h->persist_slots = 0;
    }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_ticks), 2 * kMaxTickPasses * sizeof(unsigned long long), hipHostMallocDefault));
    if (const char* s = std::getenv("NGICP_PERSIST")) h->persist = std::atoi(s);
    if (const char* s = std::getenv("NGICP_HEAD")) h->head = std::atoi(s);
    if (const char* s = std::getenv("NGICP_CELL_BOXES")) h->cell_boxes = std::atoi(s);
    if (const char* s = std::getenv("NGICP_TARGET_OCC")) h->target_occupancy = std::max(1.0, std::atof(s));
    if (const char* s = std::getenv("NGICP_VOXEL")) h->voxel_size = std::atof(s);
    if (const char* s = std::getenv("NGICP_CHUNK")) h->chunk_pairs = std::max(1, std::min(64, std::atoi(s)));
    if (const char* s = std::getenv("NGICP_STAGE_GROW")) h->stage_grow = std::max(0, std::min(kStageMaxGrow, std::atoi(s)));
    {
      HandleRegistry& r = registry();
      std::lock_guard<std::mutex> lock(r.m);
      r.live.push_back(h.get());
    }
    *out = h.release();
    return NGICP_OK;
  } catch (const HipError& e) {
    g_create_error = std::string("HIP error in ") + e.what + ": " + hipGetErrorString(e.code);
    (void)hipGetLastError();
    return NGICP_ERR_HIP;
  } catch (...) {
    g_create_error = "unknown error";
    return NGICP_ERR_ARG;
  }
}

int ngicp_destroy(ngicp_t* h) {
  if (!h) return NGICP_OK;
  {
    HandleRegistry& r = registry();
    std::lock_guard<std::mutex> lock(r.m);
    r.live.erase(std::remove(r.live.begin(), r.live.end(), h), r.live.end());
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->src.clear();
  h->tgt.clear();
  h->src_covs.clear();
  h->tgt_covs.clear();
  for (auto& e : h->prof_events)
    if (e) (void)hipEventDestroy(e);
  if (h->pin_state) (void)hipHostFree(h->pin_state);
  if (h->pin_final) (void)hipHostFree(h->pin_final);
  if (h->pin_ticks) (void)hipHostFree(h->pin_ticks);
  if (h->h_progress) (void)hipHostFree(h->h_progress);
  if (h->h_shard_done) (void)hipHostFree(h->h_shard_done);
  for (auto& e : h->ev_shard)
    if (e) (void)hipEventDestroy(e);
  if (h->ev_a) (void)hipEventDestroy(h->ev_a);
  if (h->ev_b) (void)hipEventDestroy(h->ev_b);
  if (h->ev_cov_a) (void)hipEventDestroy(h->ev_cov_a);
  if (h->ev_cov_b) (void)hipEventDestroy(h->ev_cov_b);
  if (h->ev_fence) (void)hipEventDestroy(h->ev_fence);
  ngk_filter_free(&h->fws);
  hipStream_t s = h->stream;
  delete h;
  if (s) (void)hipStreamDestroy(s);
  return NGICP_OK;
}

const char* ngicp_last_error(const ngicp_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ngicp_set_params(ngicp_t* h, int k, double max_corr_dist, int max_iter, double trans_eps, double rot_eps, int optimizer, int lm_max_iter, double lm_init_lambda_factor,
                     int regularization, int num_threads) {
  return guarded(h, [&] {
    if (regularization < 0 || regularization > 4) throw ArgError{NGICP_ERR_ARG, "unknown regularization method"};
    if (optimizer != 0 && optimizer != 1) throw ArgError{NGICP_ERR_ARG, "unknown optimizer"};
    h->p.k = k;
    h->p.max_corr_dist = max_corr_dist;
    h->p.max_iter = max_iter;
    h->p.trans_eps = trans_eps;
    h->p.rot_eps = rot_eps;
    h->p.optimizer = optimizer;
    h->p.lm_max_iter = lm_max_iter;
    h->p.lm_init_lambda_factor = lm_init_lambda_factor;
    h->p.regularization = regularization;
    h->p.num_threads = num_threads;
  });
}

int ngicp_set_tuning(ngicp_t* h, double voxel_size, int lanes_per_query) {
  return guarded(h, [&] {
    if (voxel_size < 0) throw ArgError{NGICP_ERR_ARG, "voxel_size must be >= 0"};
    if (lanes_per_query != 0 && lanes_per_query != 1 && lanes_per_query != 2 && lanes_per_query != 4 && lanes_per_query != 8 && lanes_per_query != 16)
      throw ArgError{NGICP_ERR_ARG, "lanes_per_query must be 0,1,2,4,8 or 16"};
    h->voxel_size = voxel_size;
  });
}

int ngicp_set_source(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, uint64_t id) {
  if (!h) return NGICP_ERR_ARG;
  const bool same = (id != 0 && h->src.present && h->src.identity == id);
  int rc = set_cloud(h, h->src, xyz, n, stride_bytes, id, true);
  if (rc == NGICP_OK && !same) h->src_covs.clear();  // impl/nano_gicp_impl.hpp:128
  return rc;
}
int ngicp_register_source(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, uint64_t id) {
  if (!h) return NGICP_ERR_ARG;
  return set_cloud(h, h->src, xyz, n, stride_bytes, id, false);  // covariances untouched (impl/nano_gicp_impl.hpp:113-118)
}
int ngicp_set_target(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, uint64_t id) {
  if (!h) return NGICP_ERR_ARG;
  const bool same = (id != 0 && h->tgt.present && h->tgt.identity == id);
  int rc = set_cloud(h, h->tgt, xyz, n, stride_bytes, id, true);
  if (rc == NGICP_OK && !same) h->tgt_covs.clear();  // :138
  if (!same) {  // the target is no longer the submap assembled by ngicp_submap_set (a recycled index object may reuse its address)
    h->submap_cloud = nullptr;
    h->submap_ids.clear();
  }
  return rc;
}
int ngicp_clear_source(ngicp_t* h) {
  return guarded(h, [&] {
    h->src.clear();
    h->src_covs.clear();
  });
}
int ngicp_clear_target(ngicp_t* h) {
  return guarded(h, [&] {
    h->tgt.clear();
    h->tgt_covs.clear();
    h->submap_cloud = nullptr;
    h->submap_ids.clear();
  });
}

int ngicp_share_source_index(ngicp_t* dst, ngicp_t* src) {
  if (!src) return NGICP_ERR_ARG;
  return guarded(dst, [&] {
    if (dst->device != src->device) return;  // different GPUs: dst uploads its own copy lazily
    if (!src->src.present || !src->src.dev) return;
    if (!dst->src.present) return;
    // adopt only when both refer to the same host cloud (the reference would otherwise rebuild the
    // tree for its own cloud at impl/nano_gicp_impl.hpp:304-306)
    const bool same = (dst->src.identity != 0 && dst->src.identity == src->src.identity) || (dst->src.host == src->src.host && dst->src.n == src->src.n);
    if (same) {
      // dst's stream waits (on the device) for the index build src has enqueued: no host synchronisation per scan
      HIP_TRY(hipEventRecord(dst->ev_fence, src->stream));
      HIP_TRY(hipStreamWaitEvent(dst->stream, dst->ev_fence, 0));
      dst->src.dev = src->src.dev;
    }
  });
}

int ngicp_swap_source_target(ngicp_t* h) {
  return guarded(h, [&] {
    std::swap(h->src, h->tgt);
    std::swap(h->src_covs, h->tgt_covs);
    h->hook_valid = 0;  // correspondences_.clear(); sq_distances_.clear();
    h->submap_cloud = nullptr;
    h->submap_ids.clear();
  });
}

int ngicp_compute_source_covs(ngicp_t* h) {
  return guarded(h, [&] { compute_covs(h, h->src, h->src_covs, "source"); });
}
int ngicp_compute_target_covs(ngicp_t* h) {
  return guarded(h, [&] { compute_covs(h, h->tgt, h->tgt_covs, "target"); });
}

int ngicp_copy_source_covs(ngicp_t* dst, ngicp_t* src) {
  if (!src) return NGICP_ERR_ARG;
  return guarded(dst, [&] {
    if (src->src_covs.n == 0) {
      dst->src_covs.clear();
      return;
    }
    if (dst->device == src->device) {
      HIP_TRY(hipEventRecord(dst->ev_fence, src->stream));  // the covariance kernel / reorder src has enqueued
      HIP_TRY(hipStreamWaitEvent(dst->stream, dst->ev_fence, 0));
      dst->src_covs = src->src_covs;  // shares the immutable device buffer
    } else {
      throw ArgError{NGICP_ERR_ARG, "copy_source_covs across devices is not supported; use get/set"};
    }
  });
}
int ngicp_clear_source_covs(ngicp_t* h) {
  return guarded(h, [&] { h->src_covs.clear(); });
}
int ngicp_clear_target_covs(ngicp_t* h) {
  return guarded(h, [&] { h->tgt_covs.clear(); });
}
int ngicp_source_covs_size(const ngicp_t* h, size_t* n) {
  if (!h || !n) return NGICP_ERR_ARG;
  *n = h->src_covs.n;
  return NGICP_OK;
}
int ngicp_target_covs_size(const ngicp_t* h, size_t* n) {
  if (!h || !n) return NGICP_ERR_ARG;
  *n = h->tgt_covs.n;
  return NGICP_OK;
}
int ngicp_get_source_covs(ngicp_t* h, double* out) {
  return guarded(h, [&] {
    if (!out) throw ArgError{NGICP_ERR_ARG, "null output"};
    get_covs(h, h->src_covs, out);
  });
}
int ngicp_get_target_covs(ngicp_t* h, double* out) {
  return guarded(h, [&] {
    if (!out) throw ArgError{NGICP_ERR_ARG, "null output"};
    get_covs(h, h->tgt_covs, out);
  });
}
int ngicp_set_source_covs(ngicp_t* h, const double* in, size_t n) {
  return guarded(h, [&] { set_covs(h, h->src, h->src_covs, in, n, "source"); });
}
int ngicp_set_target_covs(ngicp_t* h, const double* in, size_t n) {
  return guarded(h, [&] { set_covs(h, h->tgt, h->tgt_covs, in, n, "target"); });
}

int ngicp_align(ngicp_t* h, const float guess[16], float T_out[16], int* converged, int* nr_iterations, double final_hessian[36], float* aligned, size_t out_stride_bytes) {
  int rc = guarded(h, [&] {
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (aligned && (out_stride_bytes < 12 || out_stride_bytes % 4)) throw ArgError{NGICP_ERR_ARG, "bad out_stride_bytes"};
    do_align(h, guess ? guess : I, aligned, out_stride_bytes);
  });
  if (h) {
    if (T_out) std::memcpy(T_out, h->final_T, sizeof(h->final_T));
    if (converged) *converged = h->converged;
    if (nr_iterations) *nr_iterations = h->nr_iterations;
    if (final_hessian) std::memcpy(final_hessian, h->final_hessian, sizeof(h->final_hessian));
  }
  return rc;
}

int ngicp_linearize(ngicp_t* h, const double T[16], double H[36], double b[6], double* err) {
  return guarded(h, [&] {
    if (!T) throw ArgError{NGICP_ERR_ARG, "null pose"};
    LoopCtx c;
    prepare_loop(h, c);
    LmState st;
    init_state_from_pose(st, pose_from_colmajor_d(T));
    HIP_TRY(hipMemcpyAsync(h->state.p, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
      c.pa.mode = 2 | 4;
    launch_pass(h, c.pa, c.nblocks, h->stream);
    c.sa.mode = 1;
    hipLaunchKernelGGL(k_lm_solve, dim3(1), dim3(kSolveThreads), 0, h->stream, c.sa);
    HIP_TRY(hipMemcpyAsync(&st, h->state.p, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
    if (H)
      for (int r = 0; r < 6; ++r)
        for (int cc = 0; cc < 6; ++cc) H[cc * 6 + r] = st.hot.H[r * 6 + cc];
    if (b) std::memcpy(b, st.hot.b, sizeof(st.hot.b));
    if (err) *err = st.hot.y0;
    h->hook_valid = 1;
  });
}

int ngicp_compute_error(ngicp_t* h, const double T[16], double* err) {
  return guarded(h, [&] {
    if (!T) throw ArgError{NGICP_ERR_ARG, "null pose"};
    if (h->hook_valid != 1) throw ArgError{NGICP_ERR_STATE, "compute_error needs a preceding linearize"};
    LoopCtx c;
    prepare_loop(h, c);
    // keep cur / have_lin, replace the trial pose
    LmState st;
    HIP_TRY(hipMemcpy(&st, h->state.p, sizeof(st), hipMemcpyDeviceToHost));
    const Pose x = pose_from_colmajor_d(T);
    st.hot.xi = x;
    for (int r = 0; r < 3; ++r) {
      for (int cc = 0; cc < 3; ++cc) st.xi_f[r * 4 + cc] = (float)x.R[r * 3 + cc];
      st.xi_f[r * 4 + 3] = (float)x.t[r];
    }
    HIP_TRY(hipMemcpyAsync(h->state.p, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
    c.pa.mode = 1 | 4;
    launch_pass(h, c.pa, c.nblocks, h->stream);
    c.sa.mode = 2;
    hipLaunchKernelGGL(k_lm_solve, dim3(1), dim3(kSolveThreads), 0, h->stream, c.sa);
    LmState st2;
    HIP_TRY(hipMemcpyAsync(&st2, h->state.p, sizeof(st2), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
    if (err) *err = st2.hot.y0;
  });
}

int ngicp_get_correspondences(ngicp_t* h, int* corr_out, float* sqd_out) {
  return guarded(h, [&] {
    if (!corr_out) throw ArgError{NGICP_ERR_ARG, "null output"};
    if (!h->hook_valid) throw ArgError{NGICP_ERR_STATE, "no correspondences: call ngicp_linearize or ngicp_align first"};
    LmState st;
    HIP_TRY(hipMemcpy(&st, h->state.p, sizeof(st), hipMemcpyDeviceToHost));
    const size_t n = h->src.dev->n;
    h->knn_idx.ensure(n * sizeof(int));
    h->knn_d2.ensure(n * sizeof(float));
    LmState* dst = h->state.as<LmState>();
    hipLaunchKernelGGL(k_corr_to_original, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->tpt[st.hot.cur].as<float4>(), h->src.dev->qpts.as<float4>(),
                       h->src.dev->pts(), h->tgt.dev->pts(), (int)n, h->knn_idx.as<int>(), sqd_out ? h->knn_d2.as<float>() : nullptr, dst->xi_f);
    HIP_TRY(hipMemcpyAsync(corr_out, h->knn_idx.p, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (sqd_out) HIP_TRY(hipMemcpyAsync(sqd_out, h->knn_d2.p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
  });
}

int ngicp_target_knn(ngicp_t* h, const float* q, size_t nq, size_t stride, int k, int* idx, float* d2) {
  return guarded(h, [&] {
    if (!q || !idx || !d2) throw ArgError{NGICP_ERR_ARG, "null pointer"};
    if (nq == 0) return;
    if (stride < 12 || stride % 4) throw ArgError{NGICP_ERR_ARG, "bad stride"};
    ensure_slot_ready(h, h->tgt, "target");
    DeviceCloud& T = *h->tgt.dev;
    if (k <= 0) throw ArgError{NGICP_ERR_ARG, "k must be positive"};
    if (k > 32 || (size_t)k > T.n) throw ArgError{NGICP_ERR_K_TOO_LARGE, "k exceeds the cloud size or the engine limit of 32"};
    std::vector<float> packed(nq * 4);
    for (size_t i = 0; i < nq; ++i) {
      const float* p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(q) + i * stride);
      packed[i * 4 + 0] = p[0];
      packed[i * 4 + 1] = p[1];
      packed[i * 4 + 2] = p[2];
      packed[i * 4 + 3] = 1.f;
    }
    h->queries.ensure(nq * sizeof(float4));
    h->knn_idx.ensure(nq * k * sizeof(int));
    h->knn_d2.ensure(nq * k * sizeof(float));
    HIP_TRY(hipMemcpyAsync(h->queries.p, packed.data(), nq * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    const dim3 grid((unsigned)((nq + kKnnPairs - 1) / kKnnPairs)), block(kKnnBlock);  // a pair of lanes per query
    if (k <= 10)
      hipLaunchKernelGGL(k_knn_queries<10>, grid, block, 0, h->stream, T.pts(), T.cells(), T.grid, h->queries.as<float4>(), (int)nq, k,
                         h->knn_idx.as<int>(), h->knn_d2.as<float>());
    else if (k <= 20)
      hipLaunchKernelGGL(k_knn_queries<20>, grid, block, 0, h->stream, T.pts(), T.cells(), T.grid, h->queries.as<float4>(), (int)nq, k,
                         h->knn_idx.as<int>(), h->knn_d2.as<float>());
    else
      hipLaunchKernelGGL(k_knn_queries<32>, grid, block, 0, h->stream, T.pts(), T.cells(), T.grid, h->queries.as<float4>(), (int)nq, k,
                         h->knn_idx.as<int>(), h->knn_d2.as<float>());
    HIP_TRY(hipMemcpyAsync(idx, h->knn_idx.p, nq * k * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(d2, h->knn_d2.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
  });
}

int ngicp_get_lm_trace(ngicp_t* h, double* rows, size_t max_rows, size_t* n_rows) {
  return guarded(h, [&] {
    if (h->trace_host.empty() && h->trace_rows_dev) {
      h->trace_host.resize(h->trace_rows_dev * kTraceCols);
      HIP_TRY(hipMemcpy(h->trace_host.data(), h->trace.p, h->trace_host.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    const size_t n = h->trace_host.size() / kTraceCols;
    if (n_rows) *n_rows = n;
    if (rows) std::memcpy(rows, h->trace_host.data(), std::min(n, max_rows) * kTraceCols * sizeof(double));
  });
}

int ngicp_get_stats(ngicp_t* h, ngicp_stats* out) {
  if (!h || !out) return NGICP_ERR_ARG;
  if (h->cov_timing_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(h->ev_cov_b) == hipSuccess && hipEventElapsedTime(&ms, h->ev_cov_a, h->ev_cov_b) == hipSuccess) h->stats.covariance_ms = ms;
    h->cov_timing_pending = false;
  }
  h->stats.device_allocs = g_device_allocs.load(std::memory_order_relaxed);
  *out = h->stats;
  return NGICP_OK;
}
int ngicp_set_host_wait(ngicp_t* h, int mode) {
  if (!h || (mode != 0 && mode != 1)) return NGICP_ERR_ARG;
  h->host_wait = mode;
  return NGICP_OK;
}
int ngicp_set_profiling(ngicp_t* h, int on) {
  return guarded(h, [&] {
    h->profiling = on != 0;
    h->prof_stride = on > 1 ? on : 1;
    if (h->profiling && h->prof_events.empty()) {
      h->prof_events.resize(2 * 1024);
      for (auto& e : h->prof_events) HIP_TRY(hipEventCreate(&e));
    }
  });
}

// ---- point-sharded stepping ----
// One loop context per alignment (ngicp_sharded_begin); per pass two small launches around the caller's all-reduce (the pass +
// a reduce-only solver, then the solver proper on the reduced vector); no host synchronisation per pass: the `done` word of
// step k is copied to pinned memory behind an event and READ AT STEP k + kShardLag.  The lag is a constant, so every rank
// takes the same decision in the same step (a rank that stopped calling the collective earlier than its peers would hang them);
// the extra kShardLag passes after the end are no-ops (both kernels return at once when the state says done).
int ngicp_sharded_begin(ngicp_t* h, const float guess[16]) {
  return guarded(h, [&] {
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    h->shard_ctx.reset(new LoopCtx);
    LoopCtx& c = *h->shard_ctx;
    prepare_loop(h, c);
    c.pa.mode = (h->p.optimizer == NGICP_OPT_GAUSS_NEWTON) ? 2 : 3;
    if (h->prev_staged_fraction < 0.0 || h->prev_staged_fraction >= 0.12) c.pa.mode |= 32;  // (see do_align)
    LmState st;
    init_state_from_pose(st, pose_from_colmajor_f(guess ? guess : I));
    if (h->p.max_iter <= 0) st.hot.done = 1;
    h->pin_state[0] = st;
    HIP_TRY(hipMemcpyAsync(h->state.p, &h->pin_state[0], sizeof(st), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // once per alignment: the caller may step on a stream of its own
    for (int i = 0; i < kShardSlots; ++i) h->h_shard_done[i] = 0;
    h->shard_steps = 0;
    h->sharded_active = true;
    h->hook_valid = 0;
  });
}

int ngicp_sharded_pass(ngicp_t* h, double* sums32_dev, void* stream_or_null) {
  return guarded(h, [&] {
    if (!h->sharded_active || !h->shard_ctx) throw ArgError{NGICP_ERR_STATE, "ngicp_sharded_begin not called"};
    if (!sums32_dev) throw ArgError{NGICP_ERR_ARG, "null sums buffer"};
    hipStream_t s = stream_or_null ? (hipStream_t)stream_or_null : h->stream;
    LoopCtx& c = *h->shard_ctx;
    launch_pass(h, c.pa, c.nblocks, s);
    SolveArgs sa = c.sa;
    sa.mode = 3;  // reduce only: this rank's 32 sums, for the caller's all-reduce
    sa.sums_out = sums32_dev;
    sa.grp_order = nullptr;
    hipLaunchKernelGGL(k_lm_solve, dim3(1), dim3(kSolveThreads), 0, s, sa);
    HIP_TRY(hipGetLastError());
  });
}

int ngicp_sharded_step(ngicp_t* h, const double* sums32_dev, void* stream_or_null, int* done) {
  return guarded(h, [&] {
    if (!h->sharded_active || !h->shard_ctx) throw ArgError{NGICP_ERR_STATE, "ngicp_sharded_begin not called"};
    if (!sums32_dev) throw ArgError{NGICP_ERR_ARG, "null sums buffer"};
    hipStream_t s = stream_or_null ? (hipStream_t)stream_or_null : h->stream;
    LoopCtx& c = *h->shard_ctx;
    SolveArgs sa = c.sa;
    sa.mode = 0;
    sa.partials = sums32_dev;  // one pre-reduced row
    sa.nblocks = 1;
    sa.grp_order = nullptr;    // (the launch order of the pass is per rank and is left alone)
    hipLaunchKernelGGL(k_lm_solve, dim3(1), dim3(kSolveThreads), 0, s, sa);
    h->shard_stream = s;
    const long k = h->shard_steps++;
    const int slot = (int)(k % kShardSlots);
    LmState* dst = h->state.as<LmState>();
    HIP_TRY(hipMemcpyAsync(&h->h_shard_done[slot], &dst->hot.done, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(h->ev_shard[slot], s));
    HIP_TRY(hipGetLastError());
    int d = 0;
    if (k >= kShardLag) {  // the flag of step k - kShardLag: its copy finished long ago, the wait does not stall the stream
      const int old = (int)((k - kShardLag) % kShardSlots);
      HIP_TRY(hipEventSynchronize(h->ev_shard[old]));
      d = h->h_shard_done[old];
    }
    if (done) *done = d;
  });
}

int ngicp_sharded_finish(ngicp_t* h, float T_out[16], int* converged, int* nr_iterations, double final_hessian[36]) {
  return guarded(h, [&] {
    if (!h->sharded_active) throw ArgError{NGICP_ERR_STATE, "ngicp_sharded_begin not called"};
    hipStream_t s = h->shard_stream ? h->shard_stream : h->stream;  // the steps were enqueued there
    HIP_TRY(hipMemcpyAsync(&h->pin_state[1], h->state.p, sizeof(LmState), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const LmState& st = h->pin_state[1];
    pose_to_colmajor_f(st.hot.x0, h->final_T);
    h->converged = st.hot.converged;
    h->nr_iterations = st.hot.nr_iterations;
    for (int r = 0; r < 6; ++r)
      for (int cc = 0; cc < 6; ++cc) h->final_hessian[cc * 6 + r] = st.hot.final_H[r * 6 + cc];
    if (st.hot.lm_failed) std::fprintf(stderr, "lm not converged!!\n");  // impl/lsq_registration_impl.hpp:106
    if (T_out) std::memcpy(T_out, h->final_T, sizeof(h->final_T));
    if (converged) *converged = h->converged;
    if (nr_iterations) *nr_iterations = h->nr_iterations;
    if (final_hessian) std::memcpy(final_hessian, h->final_hessian, sizeof(h->final_hessian));
    h->sharded_active = false;
    h->shard_ctx.reset();
    h->shard_stream = nullptr;
  });
}

// ---- K1 sharded over ranks: every rank computes a block of the packed covariance array, the caller all-gathers ----
int ngicp_covs_shard_begin(ngicp_t* h, int which, double** covs6_dev, size_t* n_points) {
  return guarded(h, [&] {
    if ((which != 0 && which != 1) || !covs6_dev || !n_points) throw ArgError{NGICP_ERR_ARG, "bad argument"};
    Slot& slot = which ? h->tgt : h->src;
    ensure_slot_ready(h, slot, which ? "target" : "source");
    CovSet& cs = h->shard_covs[which];
    cs.data = acquire_buf(h, h->device, slot.dev->n * 6 * sizeof(double));
    cs.n = slot.dev->n;
    cs.order = slot.dev;
    HIP_TRY(hipStreamSynchronize(h->stream));  // (the index build; the caller may use a stream of its own from here on)
    *covs6_dev = cs.data->as<double>();
    *n_points = cs.n;
  });
}
int ngicp_covs_shard_compute(ngicp_t* h, int which, size_t lo, size_t hi, void* stream_or_null) {
  return guarded(h, [&] {
    if (which != 0 && which != 1) throw ArgError{NGICP_ERR_ARG, "bad argument"};
    CovSet& cs = h->shard_covs[which];
    Slot& slot = which ? h->tgt : h->src;
    if (!cs.data || cs.order.get() != slot.dev.get()) throw ArgError{NGICP_ERR_STATE, "ngicp_covs_shard_begin not called for this cloud"};
    if (lo > hi || hi > cs.n) throw ArgError{NGICP_ERR_ARG, "block outside the cloud"};
    launch_cov_range(h, *slot.dev, cs.data->as<double>(), lo, hi, stream_or_null ? (hipStream_t)stream_or_null : h->stream);
    HIP_TRY(hipGetLastError());
  });
}
int ngicp_covs_shard_commit(ngicp_t* h, int which) {
  return guarded(h, [&] {
    if (which != 0 && which != 1) throw ArgError{NGICP_ERR_ARG, "bad argument"};
    CovSet& cs = h->shard_covs[which];
    Slot& slot = which ? h->tgt : h->src;
    if (!cs.data || cs.order.get() != slot.dev.get()) throw ArgError{NGICP_ERR_STATE, "ngicp_covs_shard_begin not called for this cloud"};
    (which ? h->tgt_covs : h->src_covs) = cs;
    cs.clear();
  });
}

// ---- device-resident keyframe store + submap assembly ----
int ngicp_keyframe_add(ngicp_t* h, ngicp_t* from, int* id_out) {
  if (!from) return NGICP_ERR_ARG;
  return guarded(h, [&] {
    if (h->device != from->device) throw ArgError{NGICP_ERR_ARG, "keyframe_add across devices is not supported"};
    ensure_slot_ready(from, from->src, "source");
    // `keyframe_normals.push_back(gicp_s2s.getSourceCovariances())` (odom.cc:1174): the covariances the producer holds for its
    // source; computed now if absent, with the producer's k / regularisation (calculateSourceCovariances, odom.cc:1173)
    if (from->src_covs.n != from->src.dev->n) compute_covs(from, from->src, from->src_covs, "source");
    (void)covs_for(from, from->src_covs, from->src.dev);  // in the cloud's own sorted order
    HIP_TRY(hipStreamSynchronize(from->stream));          // the store is read on other streams later (keyframes are rare)
    h->keyframes.push_back({from->src.dev, from->src_covs.data});
    if (id_out) *id_out = (int)h->keyframes.size() - 1;
  });
}

int ngicp_keyframe_add_transformed(ngicp_t* h, ngicp_t* from, const float T_colmajor[16], int* id_out) {
  if (!from) return NGICP_ERR_ARG;
  return guarded(h, [&] {
    if (!T_colmajor) throw ArgError{NGICP_ERR_ARG, "null transform"};
    if (h->device != from->device) throw ArgError{NGICP_ERR_ARG, "keyframe_add across devices is not supported"};
    ensure_slot_ready(from, from->src, "source");
    // transformCurrentScan (odom.cc:971-974) + setInputSource(keyframe_cloud) + calculateSourceCovariances (odom.cc:1172-1173),
    // all on the device: the scan is already there as the producer's source
    DeviceCloud& S = *from->src.dev;
    const size_t n = S.n;
    from->tfinal.ensure(16 * sizeof(float));
    HIP_TRY(hipMemcpyAsync(from->tfinal.p, T_colmajor, 16 * sizeof(float), hipMemcpyHostToDevice, from->stream));
    from->unsorted.ensure(n * sizeof(float4));
    const int bbox_blocks = pick_blocks(n, 1024, 512);
    from->bbox.ensure((size_t)bbox_blocks * 8 * sizeof(float));
    hipLaunchKernelGGL(k_transform_to_unsorted, dim3(bbox_blocks), dim3(256), 0, from->stream, S.pts(), (int)n, from->tfinal.as<float>(), from->unsorted.as<float4>(),
                       from->bbox.as<float>());
    std::vector<float> bb((size_t)bbox_blocks * 8);
    HIP_TRY(hipMemcpyAsync(bb.data(), from->bbox.p, bb.size() * sizeof(float), hipMemcpyDeviceToHost, from->stream));
    HIP_TRY(hipStreamSynchronize(from->stream));
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int b = 0; b < bbox_blocks; ++b)
      for (int d = 0; d < 3; ++d) {
        mn[d] = std::min(mn[d], bb[(size_t)b * 8 + d]);
        mx[d] = std::max(mx[d], bb[(size_t)b * 8 + 3 + d]);
      }
    for (int d = 0; d < 3; ++d)
      if (!std::isfinite(mn[d]) || !std::isfinite(mx[d]) || mn[d] > mx[d]) throw ArgError{NGICP_ERR_ARG, "transform produced non-finite coordinates"};
    Slot tmp;
    tmp.present = true;
    tmp.n = n;
    tmp.dev = index_unsorted(from, n, mn, mx);
    CovSet cs;
    compute_covs(from, tmp, cs, "keyframe");
    HIP_TRY(hipStreamSynchronize(from->stream));
    h->keyframes.push_back({tmp.dev, cs.data});
    if (id_out) *id_out = (int)h->keyframes.size() - 1;
  });
}

int ngicp_keyframe_add_transformed_filtered(ngicp_t* h, ngicp_t* from, const float T_colmajor[16], float leaf, int* id_out) {
  if (!from) return NGICP_ERR_ARG;
  if (!(leaf > 0.f)) return ngicp_keyframe_add_transformed(h, from, T_colmajor, id_out);  // vf_submap_use_ == false
  return guarded(h, [&] {
    if (!T_colmajor) throw ArgError{NGICP_ERR_ARG, "null transform"};
    if (h->device != from->device) throw ArgError{NGICP_ERR_ARG, "keyframe_add across devices is not supported"};
    ensure_slot_ready(from, from->src, "source");
    // DLO's shipped configuration (cfg/params.yaml:33-35: voxelFilter.submap.use = true, res = 0.5): transformCurrentScan
    // (odom.cc:971-974), vf_submap.filter(*current_scan_t) (odom.cc:1160-1163), setInputSource(keyframe_cloud) +
    // calculateSourceCovariances (odom.cc:1172-1173) - the scan is already on the device as the producer's source, and nothing of
    // this visits the host: transform into the scan's ORIGINAL point order (the order VoxelGrid adds the points of a voxel in),
    // VoxelGrid centroids, index build, covariances with the producer's k.
    DeviceCloud& S = *from->src.dev;
    const size_t n = S.n;
    from->tfinal.ensure(16 * sizeof(float));
    HIP_TRY(hipMemcpyAsync(from->tfinal.p, T_colmajor, 16 * sizeof(float), hipMemcpyHostToDevice, from->stream));
    from->xyzi.ensure(n * sizeof(float4));
    hipLaunchKernelGGL(k_transform_sorted_to_original4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, from->stream, S.pts(), (int)n, from->tfinal.as<float>(),
                       from->xyzi.as<float4>());
    from->filt_out = nullptr;  // (the producer's filter workspace is reused: a preprocessed scan waiting there is gone)
    from->filt_n = 0;
    char err[256] = {0};
    const float4* out = nullptr;
    int m = 0;
    if (ngk_filter_cloud(from->stream, &from->fws, from->xyzi.as<float4>(), (int)n, 0, 0.f, leaf, &out, &m, err, sizeof(err))) throw ArgError{NGICP_ERR_HIP, err};
    if (m <= 0) throw ArgError{NGICP_ERR_ARG, "the voxel filter left no points"};
    from->unsorted.ensure((size_t)m * sizeof(float4));
    const int bbox_blocks = pick_blocks((size_t)m, 1024, 512);
    from->bbox.ensure((size_t)bbox_blocks * 8 * sizeof(float));
    hipLaunchKernelGGL(k_xyzi_to_unsorted, dim3(bbox_blocks), dim3(256), 0, from->stream, out, m, from->unsorted.as<float4>(), from->bbox.as<float>());
    std::vector<float> bb((size_t)bbox_blocks * 8);
    HIP_TRY(hipMemcpyAsync(bb.data(), from->bbox.p, bb.size() * sizeof(float), hipMemcpyDeviceToHost, from->stream));
    HIP_TRY(hipStreamSynchronize(from->stream));
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int b = 0; b < bbox_blocks; ++b)
      for (int d = 0; d < 3; ++d) {
        mn[d] = std::min(mn[d], bb[(size_t)b * 8 + d]);
        mx[d] = std::max(mx[d], bb[(size_t)b * 8 + 3 + d]);
      }
    for (int d = 0; d < 3; ++d)
      if (!std::isfinite(mn[d]) || !std::isfinite(mx[d]) || mn[d] > mx[d]) throw ArgError{NGICP_ERR_ARG, "transform produced non-finite coordinates"};
    Slot tmp;
    tmp.present = true;
    tmp.n = (size_t)m;
    tmp.dev = index_unsorted(from, (size_t)m, mn, mx);
    CovSet cs;
    compute_covs(from, tmp, cs, "keyframe");
    HIP_TRY(hipStreamSynchronize(from->stream));
    h->keyframes.push_back({tmp.dev, cs.data});
    if (id_out) *id_out = (int)h->keyframes.size() - 1;
  });
}

int ngicp_keyframe_count(const ngicp_t* h, size_t* n) {
  if (!h || !n) return NGICP_ERR_ARG;
  *n = h->keyframes.size();
  return NGICP_OK;
}

int ngicp_keyframe_size(const ngicp_t* h, int id, size_t* n_points) {
  if (!h || !n_points || id < 0 || (size_t)id >= h->keyframes.size()) return NGICP_ERR_ARG;
  *n_points = h->keyframes[(size_t)id].cloud->n;
  return NGICP_OK;
}

int ngicp_keyframe_clear(ngicp_t* h) {
  return guarded(h, [&] {
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->keyframes.clear();
    h->submap_ids.clear();
    h->submap_cloud = nullptr;
  });
}

int ngicp_submap_set(ngicp_t* h, const int* ids, size_t n_ids, int* changed_out) {
  return guarded(h, [&] {
    if (changed_out) *changed_out = 0;
    if (!ids || n_ids == 0) throw ArgError{NGICP_ERR_ARG, "empty keyframe list"};
    size_t total = 0;
    for (size_t i = 0; i < n_ids; ++i) {
      if (ids[i] < 0 || (size_t)ids[i] >= h->keyframes.size()) throw ArgError{NGICP_ERR_ARG, "unknown keyframe id"};
      total += h->keyframes[(size_t)ids[i]].cloud->n;
    }
    if (total > (size_t)0x7fffff00) throw ArgError{NGICP_ERR_ARG, "submap too large for int indices"};
    // `if (submap_kf_idx_curr == submap_kf_idx_prev) submap_hasChanged = false` (odom.cc:1308-1310, 827): same keyframes, and the
    // submap built from them is still the target -> nothing to do
    if (h->tgt.present && h->tgt.dev && h->tgt.dev.get() == h->submap_cloud && h->submap_ids.size() == n_ids &&
        std::equal(h->submap_ids.begin(), h->submap_ids.end(), ids) && h->tgt_covs.n == total)
      return;
    const double t0 = now_ms();
    // concatenation in keyframe order (odom.cc:1318-1325): point g = offset_k + (original index inside keyframe k)
    h->unsorted.ensure(total * sizeof(float4));
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    size_t off = 0;
    for (size_t i = 0; i < n_ids; ++i) {
      const ngicp::Keyframe& kf = h->keyframes[(size_t)ids[i]];
      const int nk = (int)kf.cloud->n;
      hipLaunchKernelGGL(k_keyframe_gather, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, h->stream, kf.cloud->pts(), nk, (int)off, h->unsorted.as<float4>());
      for (int d = 0; d < 3; ++d) {
        mn[d] = std::min(mn[d], kf.cloud->bb_min[d]);
        mx[d] = std::max(mx[d], kf.cloud->bb_max[d]);
      }
      off += (size_t)nk;
    }
    auto dc = index_unsorted(h, total, mn, mx);
    ensure_inv_perm(h, *dc);
    auto buf = acquire_buf(h, h->device, total * 6 * sizeof(double));
    off = 0;
    for (size_t i = 0; i < n_ids; ++i) {
      const ngicp::Keyframe& kf = h->keyframes[(size_t)ids[i]];
      const int nk = (int)kf.cloud->n;
      hipLaunchKernelGGL(k_keyframe_covs_scatter, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, h->stream, kf.cloud->pts(), kf.covs->as<double>(), nk, (int)off,
                         dc->inv_perm.as<int>(), buf->as<double>());
      off += (size_t)nk;
    }
    HIP_TRY(hipGetLastError());
    // setInputTarget(submap_cloud) + setTargetCovariances(submap_normals) (odom.cc:830-833)
    h->tgt.clear();
    h->tgt.present = true;
    h->tgt.n = total;
    h->tgt.dev = dc;
    h->tgt_covs.data = buf;
    h->tgt_covs.n = total;
    h->tgt_covs.order = dc;
    h->submap_ids.assign(ids, ids + n_ids);
    h->submap_cloud = dc.get();
    h->hook_valid = 0;
    h->stats.submap_ms = now_ms() - t0;
    if (changed_out) *changed_out = 1;
  });
}

int ngicp_get_target_points(ngicp_t* h, float* xyz_out, size_t out_stride_bytes, size_t* n_out) {
  return guarded(h, [&] {
    ensure_slot_ready(h, h->tgt, "target");
    const size_t n = h->tgt.dev->n;
    if (n_out) *n_out = n;
    if (!xyz_out) return;
    if (out_stride_bytes < 12 || out_stride_bytes % 4) throw ArgError{NGICP_ERR_ARG, "bad out_stride_bytes"};
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    download_transformed(h, *h->tgt.dev, I, xyz_out, out_stride_bytes);
  });
}

// ---- rigid transform of clouds ----
int ngicp_transform_source(ngicp_t* h, const float T_colmajor[16], float* xyz_out, size_t out_stride_bytes) {
  return guarded(h, [&] {
    if (!T_colmajor || !xyz_out) throw ArgError{NGICP_ERR_ARG, "null pointer"};
    if (out_stride_bytes < 12 || out_stride_bytes % 4) throw ArgError{NGICP_ERR_ARG, "bad out_stride_bytes"};
    ensure_slot_ready(h, h->src, "source");
    download_transformed(h, *h->src.dev, T_colmajor, xyz_out, out_stride_bytes);
  });
}

int ngicp_transform_cloud(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, const float T_colmajor[16], float* xyz_out, size_t out_stride_bytes) {
  return guarded(h, [&] {
    if (n == 0) return;
    if (!xyz || !T_colmajor || !xyz_out) throw ArgError{NGICP_ERR_ARG, "null pointer"};
    if (stride_bytes < 12 || stride_bytes % 4 || out_stride_bytes < 12 || out_stride_bytes % 4) throw ArgError{NGICP_ERR_ARG, "bad stride"};
    if (n > (size_t)0x7fffff00) throw ArgError{NGICP_ERR_ARG, "cloud too large for int indices"};
    const size_t raw_bytes = (n - 1) * stride_bytes + 12;
    h->raw.ensure(raw_bytes);
    h->tfinal.ensure(16 * sizeof(float));
    h->out_xyz.ensure(n * 3 * sizeof(float));
    HIP_TRY(hipMemcpyAsync(h->raw.p, xyz, raw_bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->tfinal.p, T_colmajor, 16 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_transform_raw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->raw.as<unsigned char>(), stride_bytes, (int)n, h->tfinal.as<float>(),
                       h->out_xyz.as<float>());
    download_xyz(h, n, xyz_out, out_stride_bytes);
  });
}

// ---- test hook: ngicp_math.h on the device ----
int ngicp_math_selftest(ngicp_t* h, int which, const double* in, size_t n_problems, double* out) {
  return guarded(h, [&] {
    static const int kIn[4] = {3, 42, 6, 6}, kOut[4] = {9, 6, 12, 6};
    if (which < 0 || which > 3 || !in || !out) throw ArgError{NGICP_ERR_ARG, "bad arguments"};
    if (n_problems == 0) return;
    DevBuf di, dout;
    di.ensure(n_problems * kIn[which] * sizeof(double));
    dout.ensure(n_problems * kOut[which] * sizeof(double));
    HIP_TRY(hipMemcpyAsync(di.p, in, n_problems * kIn[which] * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_math_selftest, dim3((unsigned)((n_problems + 63) / 64)), dim3(64), 0, h->stream, which, di.as<double>(), (int)n_problems, dout.as<double>());
    HIP_TRY(hipMemcpyAsync(out, dout.p, n_problems * kOut[which] * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
  });
}

// ---- measurement: device stream copy ----
int ngicp_measure_copy_bandwidth(ngicp_t* h, size_t bytes, int reps, double* gbps_out) {
  return guarded(h, [&] {
    if (!gbps_out || bytes < 4096 || reps <= 0) throw ArgError{NGICP_ERR_ARG, "bad arguments"};
    const size_t n16 = bytes / 16;
    DevBuf a, b;
    a.ensure(n16 * 16);
    b.ensure(n16 * 16);
    HIP_TRY(hipMemsetAsync(a.p, 1, n16 * 16, h->stream));
    const unsigned blocks = (unsigned)((n16 + 1023) / 1024);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, h->stream, a.as<float4>(), b.as<float4>(), n16);
    HIP_TRY(hipEventRecord(h->ev_a, h->stream));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, h->stream, a.as<float4>(), b.as<float4>(), n16);
    HIP_TRY(hipEventRecord(h->ev_b, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev_b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    HIP_TRY(hipGetLastError());
    *gbps_out = 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;  // read + write
  });
}

}  // extern "C"

// ---- scan preprocessing and map accumulation + voxel filter ----
namespace {
// host cloud (strided xyz [+ intensity]) -> device float4 {x, y, z, intensity} at dst[0..n)
void upload_xyzi(ngicp* h, const float* pts, size_t n, size_t stride, long intensity_off, float4* dst) {
  if (stride < 12 || stride % 4) throw ArgError{NGICP_ERR_ARG, "stride_bytes must be a multiple of 4 and >= 12"};
  if (intensity_off >= 0 && ((size_t)intensity_off + 4 > stride || intensity_off % 4)) throw ArgError{NGICP_ERR_ARG, "intensity offset outside the point stride"};
  if (n > (size_t)0x7fffff00) throw ArgError{NGICP_ERR_ARG, "cloud too large for int indices"};
  const size_t raw_bytes = (n - 1) * stride + (intensity_off >= 0 ? std::max((size_t)12, (size_t)intensity_off + 4) : 12);
  h->raw.ensure(raw_bytes);
  HIP_TRY(hipMemcpyAsync(h->raw.p, pts, raw_bytes, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_unpack_xyzi, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->raw.as<unsigned char>(), stride, intensity_off, (int)n, dst);
}
void download_xyzi(ngicp* h, const float4* src, size_t n, float* out) {
  if (n == 0) return;
  HIP_TRY(hipMemcpyAsync(out, src, n * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
}
}  // namespace

extern "C" {

int ngicp_preprocess_scan(ngicp_t* h, const float* pts, size_t n, size_t stride_bytes, long intensity_offset_bytes, int remove_nan, float crop_half_extent,
                          float voxel_leaf, float* out_xyzi, size_t out_capacity, size_t* n_out) {
  return guarded(h, [&] {
    if (n_out) *n_out = 0;
    h->filt_out = nullptr;
    h->filt_n = 0;
    if (n == 0) return;
    if (!pts) throw ArgError{NGICP_ERR_ARG, "null cloud pointer"};
    h->xyzi.ensure(n * sizeof(float4));
    upload_xyzi(h, pts, n, stride_bytes, intensity_offset_bytes, h->xyzi.as<float4>());
    char err[256] = {0};
    const float4* out = nullptr;
    int m = 0;
    if (ngk_filter_cloud(h->stream, &h->fws, h->xyzi.as<float4>(), (int)n, remove_nan, crop_half_extent, voxel_leaf, &out, &m, err, sizeof(err)))
      throw ArgError{NGICP_ERR_HIP, err};
    h->filt_out = out;
    h->filt_n = m;
    if (n_out) *n_out = (size_t)m;
    if (out_xyzi) {
      if ((size_t)m > out_capacity) throw ArgError{NGICP_ERR_ARG, "output buffer too small for the filtered cloud"};
      download_xyzi(h, out, (size_t)m, out_xyzi);
    }
  });
}

int ngicp_set_source_preprocessed(ngicp_t* h, uint64_t host_identity) {
  int rc = guarded(h, [&] {
    if (!h->filt_out || h->filt_n <= 0) throw ArgError{NGICP_ERR_STATE, "no preprocessed cloud: call ngicp_preprocess_scan first"};
    const size_t n = (size_t)h->filt_n;
    h->unsorted.ensure(n * sizeof(float4));
    const int bbox_blocks = pick_blocks(n, 1024, 512);
    h->bbox.ensure((size_t)bbox_blocks * 8 * sizeof(float));
    hipLaunchKernelGGL(k_xyzi_to_unsorted, dim3(bbox_blocks), dim3(256), 0, h->stream, h->filt_out, (int)n, h->unsorted.as<float4>(), h->bbox.as<float>());
    std::vector<float> bb((size_t)bbox_blocks * 8);
    HIP_TRY(hipMemcpyAsync(bb.data(), h->bbox.p, bb.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int b = 0; b < bbox_blocks; ++b)
      for (int d = 0; d < 3; ++d) {
        mn[d] = std::min(mn[d], bb[(size_t)b * 8 + d]);
        mx[d] = std::max(mx[d], bb[(size_t)b * 8 + 3 + d]);
      }
    for (int d = 0; d < 3; ++d)
      if (!std::isfinite(mn[d]) || !std::isfinite(mx[d]) || mn[d] > mx[d]) throw ArgError{NGICP_ERR_ARG, "preprocessed cloud contains non-finite coordinates (filter it with remove_nan)"};
    auto dc = index_unsorted(h, n, mn, mx);
    h->src.clear();  // setInputSource (impl/nano_gicp_impl.hpp:121-129) with a cloud that is already on the device
    h->src.present = true;
    h->src.n = n;
    h->src.identity = host_identity;
    h->src.dev = dc;
    h->src_covs.clear();
  });
  return rc;
}

int ngicp_map_add(ngicp_t* h, const float* pts, size_t n, size_t stride_bytes, long intensity_offset_bytes) {
  return guarded(h, [&] {
    if (n == 0) return;
    if (!pts) throw ArgError{NGICP_ERR_ARG, "null cloud pointer"};
    if (h->map_n + n > (size_t)0x7fffff00) throw ArgError{NGICP_ERR_ARG, "map too large for int indices"};
    if ((h->map_n + n) * sizeof(float4) > h->map_pts.cap) {  // grow, keeping what is there (`*dlo_map += *keyframe`, map.cc:129)
      DevBuf bigger;
      bigger.ensure((h->map_n + n) * 2 * sizeof(float4));
      if (h->map_n) HIP_TRY(hipMemcpyAsync(bigger.p, h->map_pts.p, h->map_n * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      std::swap(bigger.p, h->map_pts.p);
      std::swap(bigger.cap, h->map_pts.cap);
    }
    upload_xyzi(h, pts, n, stride_bytes, intensity_offset_bytes, h->map_pts.as<float4>() + h->map_n);
    h->map_n += n;
    HIP_TRY(hipGetLastError());
  });
}

int ngicp_map_voxel_filter(ngicp_t* h, float leaf, size_t* n_out) {
  return guarded(h, [&] {
    if (n_out) *n_out = h->map_n;
    if (h->map_n == 0 || !(leaf > 0.f)) return;
    char err[256] = {0};
    const float4* out = nullptr;
    int m = 0;
    // The filter workspace is shared with ngicp_preprocess_scan: whatever that call left there (h->filt_out points into it) is
    // overwritten or reallocated now, so a later ngicp_set_source_preprocessed must find nothing rather than stale memory.
    h->filt_out = nullptr;
    h->filt_n = 0;
    // voxelgrid.setInputCloud(dlo_map); voxelgrid.filter(*dlo_map)  (map.cc:102-104): the map is replaced by its centroids
    if (ngk_filter_cloud(h->stream, &h->fws, h->map_pts.as<float4>(), (int)h->map_n, 0, 0.f, leaf, &out, &m, err, sizeof(err))) throw ArgError{NGICP_ERR_HIP, err};
    if (out != h->map_pts.as<float4>()) {
      HIP_TRY(hipMemcpyAsync(h->map_pts.p, out, (size_t)m * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->map_n = (size_t)m;
    if (n_out) *n_out = h->map_n;
  });
}

int ngicp_map_size(const ngicp_t* h, size_t* n) {
  if (!h || !n) return NGICP_ERR_ARG;
  *n = h->map_n;
  return NGICP_OK;
}

int ngicp_map_get(ngicp_t* h, float* out_xyzi, size_t out_capacity) {
  return guarded(h, [&] {
    if (!out_xyzi) throw ArgError{NGICP_ERR_ARG, "null output"};
    if (h->map_n > out_capacity) throw ArgError{NGICP_ERR_ARG, "output buffer too small for the map"};
    download_xyzi(h, h->map_pts.as<float4>(), h->map_n, out_xyzi);
  });
}

int ngicp_map_clear(ngicp_t* h) {
  return guarded(h, [&] { h->map_n = 0; });
}

}  // extern "C"
