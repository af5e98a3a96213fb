/* =============================================================================
 * ngicp.h — C ABI of the MI355X-native NanoGICP scan-matching engine.
 *
 * One opaque handle per nano_gicp::NanoGICP<PointXYZI,PointXYZI> instance
 * (DLO holds two: include/dlo/odom.h:119-120).  Plain pointers and sizes only:
 * no C++/torch/PCL/Eigen types cross this boundary.  Paths below are relative to
 * the reference tree (/root/reference/).
 *
 * Conventions
 *  - every entry returns an int status: 0 = OK, <0 = error (ngicp_last_error()
 *    gives the text); nothing throws or aborts across the boundary;
 *  - a handle is single-caller at a time but may be called from any host thread
 *    (DLO's AsyncSpinner(0): src/dlo/odom_node.cc:27); each entry selects the
 *    handle's device and works on the handle's own HIP stream;
 *  - clouds are given as a pointer to the first float of the first point plus a
 *    byte stride (32 for pcl::PointXYZI, include/dlo/dlo.h:50; 12 for packed xyz);
 *    only x,y,z are read.  `host_identity` is the caller's stand-in for the
 *    reference's shared_ptr identity test (include/nano_gicp/impl/nano_gicp_impl.hpp:
 *    114,122,133): a call with the identity already set on that slot is a no-op;
 *    0 means "no identity, always re-upload";
 *  - 4x4 matrices are column-major (Eigen default); covariances travel as the
 *    reference's std::vector<Eigen::Matrix4d> memory image: N x 16 doubles,
 *    column-major, 3x3 block + zero 4th row/column, in the cloud's ORIGINAL point
 *    order (the engine keeps its own cell-sorted order internally).
 * ============================================================================= */
#ifndef NGICP_H
#define NGICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngicp ngicp_t;

/* status codes */
#define NGICP_OK 0
#define NGICP_ERR_HIP (-1)        /* a HIP runtime call failed (no device, OOM, ...) */
#define NGICP_ERR_ARG (-2)        /* bad argument */
#define NGICP_ERR_STATE (-3)      /* call sequence error (e.g. align() without target) */
#define NGICP_ERR_K_TOO_LARGE (-4)/* k > cloud size or k > 32: undefined in the reference, explicit error here */

/* regularisation (include/nano_gicp/gicp/gicp_settings.hpp:47) */
#define NGICP_REG_NONE 0
#define NGICP_REG_MIN_EIG 1
#define NGICP_REG_NORMALIZED_MIN_EIG 2
#define NGICP_REG_PLANE 3
#define NGICP_REG_FROBENIUS 4

/* optimiser (include/nano_gicp/lsq_registration.hpp:54) */
#define NGICP_OPT_GAUSS_NEWTON 0
#define NGICP_OPT_LEVENBERG_MARQUARDT 1

/* --- lifetime ------------------------------------------------------------- */
/* NanoGICP::NanoGICP()  impl/nano_gicp_impl.hpp:50-64 (+ LsqRegistration ctor
 * impl/lsq_registration_impl.hpp:50-63): k=20, PLANE, LM, max_iter 64, rot_eps 2e-3,
 * trans_eps 5e-4, lm_max 10, lambda factor 1e-9, corr dist FLT_MAX. */
int ngicp_create(int device, ngicp_t** out);
int ngicp_destroy(ngicp_t* h);
const char* ngicp_last_error(const ngicp_t* h); /* h may be NULL: last create() error */
const char* ngicp_version(void);

/* --- parameters ----------------------------------------------------------- */
/* setCorrespondenceRandomness impl/nano_gicp_impl.hpp:81-83; setMaxCorrespondenceDistance /
 * setMaximumIterations / setTransformationEpsilon (PCL base; consumed at impl/nano_gicp_impl.hpp:195,
 * impl/lsq_registration_impl.hpp:101,124); setRotationEpsilon / setInitialLambdaFactor
 * impl/lsq_registration_impl.hpp:69-76; setRegularizationMethod impl/nano_gicp_impl.hpp:86-88;
 * setNumThreads impl/nano_gicp_impl.hpp:70-78 (accepted, meaningless on the GPU). */
int ngicp_set_params(ngicp_t* h, int k, double max_corr_dist, int max_iter, double trans_eps, double rot_eps,
                     int optimizer, int lm_max_iter, double lm_init_lambda_factor, int regularization, int num_threads);
/* engine knob with no reference counterpart: voxel edge of the search grid in metres (0 = automatic: sized so that a
 * point sees ~24 points in its own cell).  A pure performance knob: the search is exact for any grid, so results do not
 * depend on it beyond the summation order.  `lanes_per_query` is accepted for ABI stability (0, 1, 2, 4, 8 or 16) and
 * ignored: the per-iteration kernel is built for 32-query batches with 2 lanes per query. */
int ngicp_set_tuning(ngicp_t* h, double voxel_size, int lanes_per_query);
/* How the calling thread waits for the device inside ngicp_align(): 0 (default) polls the solver's progress word without giving
 * the core up (lowest latency: a 100k -> 500k alignment is ~1 ms); 1 yields the core between polls (sched_yield: for hosts
 * that run other work on the same core - the node's other callbacks in DLO's AsyncSpinner, src/dlo/odom_node.cc:27).
 * Results do not depend on it. */
int ngicp_set_host_wait(ngicp_t* h, int mode);

/* --- clouds --------------------------------------------------------------- */
/* setInputSource impl/nano_gicp_impl.hpp:121-129: store cloud, (re)build index, clear source covs. */
int ngicp_set_source(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, uint64_t host_identity);
/* registerInputSource impl/nano_gicp_impl.hpp:113-118: store cloud only; covariances untouched.
 * The pointer must stay valid until the next set/register/clear of the source (the reference
 * holds a shared_ptr): upload is deferred so that ngicp_share_source_index can avoid it. */
int ngicp_register_source(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, uint64_t host_identity);
/* setInputTarget impl/nano_gicp_impl.hpp:132-139 */
int ngicp_set_target(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, uint64_t host_identity);
/* clearSource / clearTarget impl/nano_gicp_impl.hpp:101-110 */
int ngicp_clear_source(ngicp_t* h);
int ngicp_clear_target(ngicp_t* h);
/* `gicp.source_kdtree_ = gicp_s2s.source_kdtree_;`  src/dlo/odom.cc:525 — dst adopts src's
 * device-resident source cloud + index when it refers to the same host cloud. */
int ngicp_share_source_index(ngicp_t* dst, ngicp_t* src);
/* swapSourceAndTarget impl/nano_gicp_impl.hpp:91-98 */
int ngicp_swap_source_target(ngicp_t* h);

/* --- covariances ---------------------------------------------------------- */
/* calculateSourceCovariances / calculateTargetCovariances impl/nano_gicp_impl.hpp:152-159,300-357 */
int ngicp_compute_source_covs(ngicp_t* h);
int ngicp_compute_target_covs(ngicp_t* h);
/* `gicp.source_covs_ = gicp_s2s.source_covs_;` src/dlo/odom.cc:815 (device-to-device) */
int ngicp_copy_source_covs(ngicp_t* dst, ngicp_t* src);
/* `gicp.source_covs_.clear();` src/dlo/odom.cc:526 */
int ngicp_clear_source_covs(ngicp_t* h);
int ngicp_clear_target_covs(ngicp_t* h);
/* source_covs_.size() / target_covs_.size() */
int ngicp_source_covs_size(const ngicp_t* h, size_t* n);
int ngicp_target_covs_size(const ngicp_t* h, size_t* n);
/* getSourceCovariances / getTargetCovariances include/nano_gicp/nano_gicp.hpp:100-106 */
int ngicp_get_source_covs(ngicp_t* h, double* out_n16);
int ngicp_get_target_covs(ngicp_t* h, double* out_n16);
/* setSourceCovariances / setTargetCovariances impl/nano_gicp_impl.hpp:142-149 */
int ngicp_set_source_covs(ngicp_t* h, const double* in_n16, size_t n);
int ngicp_set_target_covs(ngicp_t* h, const double* in_n16, size_t n);

/* --- registration --------------------------------------------------------- */
/* pcl::Registration::align(output, guess) -> NanoGICP::computeTransformation
 * impl/nano_gicp_impl.hpp:162-171 -> LsqRegistration::computeTransformation
 * impl/lsq_registration_impl.hpp:89-115.  Outputs: final_transformation_ (float 4x4),
 * converged_, nr_iterations_ (index of the last iteration), final_hessian_ (6x6), and — when
 * aligned_xyz_or_null != NULL — the source cloud transformed by the float matrix
 * (pcl::transformPointCloud, impl/lsq_registration_impl.hpp:114), xyz written at out_stride_bytes. */
int ngicp_align(ngicp_t* h, const float guess_colmajor[16], float T_out_colmajor[16], int* converged, int* nr_iterations,
                double final_hessian_colmajor[36], float* aligned_xyz_or_null, size_t out_stride_bytes);

/* --- parity / test hooks --------------------------------------------------- */
/* NanoGICP::linearize impl/nano_gicp_impl.hpp:214-270 (includes update_correspondences :174-211) */
int ngicp_linearize(ngicp_t* h, const double T_colmajor[16], double H_colmajor[36], double b[6], double* err);
/* NanoGICP::compute_error impl/nano_gicp_impl.hpp:273-296 (stale correspondences of the last linearize) */
int ngicp_compute_error(ngicp_t* h, const double T_colmajor[16], double* err);
/* correspondences_ / sq_distances_ of the last linearize, mapped back to ORIGINAL source/target
 * point indices (-1 = gated out). sq_dist may be NULL.  Also valid after ngicp_align: then the indices are those of the
 * alignment's last linearisation (the reference's correspondences_ after align); sq_dist is only meaningful after
 * ngicp_linearize. */
int ngicp_get_correspondences(ngicp_t* h, int* corr_n, float* sq_dist_n_or_null);
/* exact k-NN of arbitrary query points in the TARGET cloud (KdTreeFLANN::nearestKSearch,
 * include/nano_gicp/nanoflann.hpp:141-152): original target indices + float squared distances, ascending. */
int ngicp_target_knn(ngicp_t* h, const float* queries_xyz, size_t nq, size_t stride_bytes, int k, int* idx_nq_k, float* sqd_nq_k);
/* LM trace of the last align(): rows of 8 doubles {outer, trial, y0, yi, rho, lambda, |d|, accepted}
 * (the columns setDebugPrint prints, impl/lsq_registration_impl.hpp:183-189). */
int ngicp_get_lm_trace(ngicp_t* h, double* rows8_or_null, size_t max_rows, size_t* n_rows);

/* The small FP64 routines of the engine evaluated on the device, one problem per thread (unit-test hook): which = 0 so3_exp
 * (gicp/so3.hpp:99-118 followed by Quaternion::toRotationMatrix; in: 3 doubles, out: R row-major 9), 1 the 6x6 LDLT solve that
 * stands in for Eigen::LDLT (impl/lsq_registration_impl.hpp:147-148,172-173; in: A row-major 36 + rhs 6, out: 6), 2 the
 * symmetric 3x3 eigen-decomposition that stands in for JacobiSVD (impl/nano_gicp_impl.hpp:332; in: {xx,xy,xz,yy,yz,zz}, out:
 * w 3 + V row-major 9), 3 the symmetric 3x3 inverse (impl/nano_gicp_impl.hpp:205-209; in 6, out 6). */
int ngicp_math_selftest(ngicp_t* h, int which, const double* in, size_t n_problems, double* out);

/* --- measurement ----------------------------------------------------------- */
typedef struct ngicp_stats {
  double align_ms;          /* host wall time of the last ngicp_align() */
  double loop_ms;           /* device time (HIP events on the handle's stream) of the iteration loop */
  double pass_ms_total;     /* device time (HIP events on the handle's stream) summed over the k_gicp_pass launches of the
                               last align that did work; only collected while ngicp_set_profiling(h, 1) */
  int passes;               /* per-iteration kernels launched that did work (= linearisations incl. the speculative last one) */
  int outer_iterations;     /* nr_iterations_ + 1 */
  int lm_trials;            /* LM trials evaluated */
  double mean_candidates;   /* C-bar: target points distance-tested per source point per pass */
  double valid_fraction;    /* fraction of source points with a correspondence in the last pass */
  double index_build_ms;    /* device time of the last index build on this handle */
  double covariance_ms;     /* device time of the last covariance computation */
  double upload_ms;         /* host wall time of the last cloud upload */
  double voxel_size;        /* target grid voxel edge in use */
  int grid_dims[3];
  int lanes_per_query;
  int passes_timed;         /* launches covered by pass_ms_total */
  long long n_src, n_tgt;   /* cloud sizes of the last align */
  double staged_fraction;   /* fraction of queries served through the LDS row index of their batch region */
  double submap_ms;         /* host wall time of the last ngicp_submap_set() that rebuilt the target (enqueue + index build) */
  long long device_allocs;  /* hipMalloc calls made by this process's engine buffers so far (a call that grows a buffer in the middle of
                               a frame shows up as a latency outlier: two readings around a call attribute it) */
  long long host_wait_spins; /* polls of the solver's progress word during the last ngicp_align() (busy or yielding, see below) */
} ngicp_stats;
int ngicp_get_stats(ngicp_t* h, ngicp_stats* out);
/* HIP-event timing of the k_gicp_pass launches inside align (two event records per timed launch; off by default).
   on = 1: every launch; on = N > 1: every N-th launch (an event between two kernels costs a few microseconds of stream time) */
int ngicp_set_profiling(ngicp_t* h, int on);

/* --- point-sharded multi-GPU stepping ------------------------ */
/* One alignment whose SOURCE points are split over ranks (each rank's handle holds its block of the source, the whole target
 * and both covariance sets).  Per pass: ngicp_sharded_pass runs the fused pass over this rank's block at the engine's current
 * trial pose and leaves 32 doubles {H upper-tri 21, b 6, y0, yi, candidates tested, valid correspondences, listed queries} in
 * device memory `sums32_dev`; the caller all-reduces them (RCCL through torch.distributed: 256 B, latency-bound);
 * ngicp_sharded_step feeds the reduced sums to the LM state machine, which advances identically on every rank (what is
 * reduced is the reference's own per-thread partial sum, impl/nano_gicp_impl.hpp:260-267).  Nothing synchronises the host per
 * pass: *done reports the state as of TWO steps earlier (a constant lag, so that every rank leaves the loop in the same step -
 * a rank that stopped calling the collective before its peers would hang them); the two passes after the end do nothing.
 * All calls of one alignment must use the same stream (NULL: the handle's own). */
int ngicp_sharded_begin(ngicp_t* h, const float guess_colmajor[16]);
int ngicp_sharded_pass(ngicp_t* h, double* sums32_dev, void* hip_stream_or_null);
int ngicp_sharded_step(ngicp_t* h, const double* sums32_dev, void* hip_stream_or_null, int* done);
int ngicp_sharded_finish(ngicp_t* h, float T_out_colmajor[16], int* converged, int* nr_iterations, double final_hessian_colmajor[36]);
/* K1 sharded the same way ("K1 shards the same way with an all-gather of the packed covariances"; the loop being split is
 * impl/nano_gicp_impl.hpp:309-354).  Every rank holds the whole cloud (the k-NN of a point looks at all of it) and the same index
 * (the build is deterministic), so the packed covariance array [n][6] FP64 has the same layout on every rank: rank r computes the
 * rows of the points at sorted positions [lo, hi) of the source (which = 0) or target (which = 1) cloud in place, the caller
 * all-gathers the blocks (RCCL through torch.distributed on the device pointer *covs6_dev, n * 6 doubles) and commits the set.
 *   ngicp_covs_shard_begin    allocates the set (uncommitted: align() will not use it) and returns its device pointer and n
 *   ngicp_covs_shard_compute  computes the block [lo, hi) (sorted positions) on the handle's stream, or on `stream` if given
 *   ngicp_covs_shard_commit   the set becomes the cloud's covariances (as after ngicp_compute_*_covs) */
int ngicp_covs_shard_begin(ngicp_t* h, int which, double** covs6_dev, size_t* n_points);
int ngicp_covs_shard_compute(ngicp_t* h, int which, size_t lo, size_t hi, void* hip_stream_or_null);
int ngicp_covs_shard_commit(ngicp_t* h, int which);

/* --- device-resident keyframe store + submap assembly ------------ */
/* DLO keeps every keyframe twice on the host: its cloud (`keyframes`, src/dlo/odom.cc:1166) and its covariances
 * (`keyframe_normals`, odom.cc:1172-1174, computed by gicp_s2s used as a covariance service), and on every change of the
 * selected keyframe set concatenates both (odom.cc:1318-1325) and hands them to gicp (odom.cc:830-833): a re-upload, a
 * re-index and a 128 B/point covariance image per change.  Here the keyframes stay on the device:
 *   ngicp_keyframe_add(h, from, &id)   replaces odom.cc:1174: h's store adopts `from`'s current SOURCE cloud (already uploaded
 *                                      and indexed by setInputSource, odom.cc:1172) and its source covariances (computed with
 *                                      `from`'s k / regularisation if not yet present, odom.cc:1173).  No copy.  `from` may be h.
 *   ngicp_keyframe_add_transformed     replaces odom.cc:971-974 + 1166-1174 when the submap voxel filter is off: the keyframe is
 *                                      `from`'s current source cloud (the scan, already on the device) transformed by the float
 *                                      matrix T (pcl::transformPointCloud), indexed and given covariances with `from`'s k, all
 *                                      on the device.  `from`'s own source slot is left untouched.
 *   ngicp_submap_set(h, ids, n, &chg)  replaces odom.cc:1318-1325 + 830-833: target := the keyframes ids[0..n) concatenated in
 *                                      that order (point g = offset_k + original index inside keyframe k, exactly the host
 *                                      concatenation), target covariances := their covariances likewise; one index build, no
 *                                      host traffic.  A call with the id list the current target was built from is a no-op
 *                                      (*chg = 0), like the submap_hasChanged test (odom.cc:827,1308).
 * Keyframe ids are dense, in insertion order (the index DLO uses for `keyframes[k]`). */
int ngicp_keyframe_add(ngicp_t* h, ngicp_t* from, int* id_out);
int ngicp_keyframe_add_transformed(ngicp_t* h, ngicp_t* from, const float T_colmajor[16], int* id_out);
/* The same for DLO's SHIPPED configuration (cfg/params.yaml:33-35: voxelFilter.submap.use = true, res = 0.5), where the transformed
 * scan is voxel-filtered BEFORE it becomes a keyframe (src/dlo/odom.cc:1160-1163): replaces odom.cc:971-974 + 1160-1174.  The
 * keyframe is pcl::VoxelGrid(leaf) of `from`'s current source cloud transformed by T - transform (in the scan's original point
 * order), centroids, index build and covariances all on the device.  leaf <= 0 is ngicp_keyframe_add_transformed. */
int ngicp_keyframe_add_transformed_filtered(ngicp_t* h, ngicp_t* from, const float T_colmajor[16], float leaf, int* id_out);
int ngicp_keyframe_count(const ngicp_t* h, size_t* n);
int ngicp_keyframe_size(const ngicp_t* h, int id, size_t* n_points);
int ngicp_keyframe_clear(ngicp_t* h);
int ngicp_submap_set(ngicp_t* h, const int* ids, size_t n_ids, int* changed_out_or_null);
/* the target cloud as the engine holds it, in ORIGINAL point order (for a device-assembled submap: the concatenation).
 * xyz_out may be NULL to query the size only. */
int ngicp_get_target_points(ngicp_t* h, float* xyz_out_or_null, size_t out_stride_bytes, size_t* n_out_or_null);

/* --- rigid transform of clouds ------------------------------------ */
/* pcl::transformPointCloud(in, out, Eigen::Matrix4f) — impl/lsq_registration_impl.hpp:114, src/dlo/odom.cc:484,971-974.
 * ngicp_transform_source: the handle's current source cloud (already on the device) -> host, original point order;
 * ngicp_transform_cloud: any host cloud -> host (upload, transform, download).  Float arithmetic in PCL's order, no FMA. */
int ngicp_transform_source(ngicp_t* h, const float T_colmajor[16], float* xyz_out, size_t out_stride_bytes);
int ngicp_transform_cloud(ngicp_t* h, const float* xyz, size_t n, size_t stride_bytes, const float T_colmajor[16], float* xyz_out, size_t out_stride_bytes);

/* --- scan preprocessing ---------------------------------------------- */
/* dlo::OdomNode::preprocessPoints (src/dlo/odom.cc:443-465, configured at :122-127): pcl::removeNaNFromPointCloud, then
 * pcl::CropBox with setNegative(true) and min/max = -/+crop_half_extent (drops the points inside the cube around the
 * sensor), then pcl::VoxelGrid with a cubic leaf (one centroid of x, y, z AND intensity per occupied voxel, in ascending
 * voxel index).  Each stage is optional (remove_nan = 0, crop_half_extent <= 0, voxel_leaf <= 0).  Input: strided points,
 * xyz at byte 0 and (optionally) a float intensity at intensity_offset_bytes (16 for pcl::PointXYZI; -1: none).  Output:
 * 16 bytes per point {x, y, z, intensity}, written to out_xyzi (capacity in points; may be NULL) and kept on the device.
 * PCL's sources are not under /root/reference: the rules are restated from memory (see csrc/ngicp_filters.hip).
 * ngicp_set_source_preprocessed makes the filtered cloud (still on the device) the handle's source: setInputSource
 * (odom.cc:519) without the download / upload pair. */
int ngicp_preprocess_scan(ngicp_t* h, const float* pts, size_t n, size_t stride_bytes, long intensity_offset_bytes, int remove_nan, float crop_half_extent,
                          float voxel_leaf, float* out_xyzi_or_null, size_t out_capacity, size_t* n_out);
int ngicp_set_source_preprocessed(ngicp_t* h, uint64_t host_identity);

/* --- map accumulation + voxel filter ---------------------------------- */
/* dlo::MapNode (src/dlo/map.cc:100-131): `*dlo_map += *keyframe` per keyframe (ngicp_map_add: the keyframe is appended to
 * a device-resident map), and on the publish timer `voxelgrid.filter(*dlo_map)` with a cubic leaf (ngicp_map_voxel_filter:
 * the map is replaced by its voxel centroids, same rules as above); ngicp_map_get downloads it for publishing
 * ({x, y, z, intensity}, 16 bytes per point). */
int ngicp_map_add(ngicp_t* h, const float* pts, size_t n, size_t stride_bytes, long intensity_offset_bytes);
int ngicp_map_voxel_filter(ngicp_t* h, float leaf, size_t* n_out_or_null);
int ngicp_map_size(const ngicp_t* h, size_t* n);
int ngicp_map_get(ngicp_t* h, float* out_xyzi, size_t out_capacity);
int ngicp_map_clear(ngicp_t* h);

/* --- measurement: device stream copy -------------------------------- */
/* float4 grid-stride copy of `bytes` bytes, `reps` times on the handle's stream, HIP-event timed: (read + write) GB/s. */
int ngicp_measure_copy_bandwidth(ngicp_t* h, size_t bytes, int reps, double* gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* NGICP_H */
