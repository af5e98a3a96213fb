// Cloud-level kernels around the registration path: submap assembly from device-resident keyframes and the
// rigid transform of clouds.
//   keyframes + submap   /root/reference/src/dlo/odom.cc:1166-1174 (keyframe cloud + covariances), :1318-1325 (concatenation of
//                        the selected keyframes' clouds and covariance vectors, in keyframe order), :827-834 (hand-over to gicp)
//   rigid transform      pcl::transformPointCloud with a float matrix: impl/lsq_registration_impl.hpp:114, odom.cc:484,971-974
#pragma once
#include "ngicp_grid.h"
#include "ngicp_math.h"

namespace ngk {

// One keyframe's cell-sorted points -> its slice of the submap's staging array, in the point order the host concatenation would
// have produced: element (offset + original index) = {x, y, z, bitcast(offset + original index)}.  The index build that follows
// sees exactly the cloud `*submap_cloud_ += *keyframes[k]` (odom.cc:1321) would have uploaded.
__global__ void __launch_bounds__(256) k_keyframe_gather(const float4* __restrict__ kf_sorted, int n, int offset, float4* __restrict__ unsorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = kf_sorted[i];
  const int g = offset + __float_as_int(p.w);
  unsorted[g] = make_float4(p.x, p.y, p.z, __int_as_float(g));
}

// One keyframe's packed covariances (its own sorted order) -> the submap's covariance set (the submap's sorted order):
// the device form of `submap_normals.insert(end, keyframe_normals[k].begin(), keyframe_normals[k].end())` (odom.cc:1324).
__global__ void __launch_bounds__(256) k_keyframe_covs_scatter(const float4* __restrict__ kf_sorted, const double* __restrict__ kf_covs6, int n, int offset,
                                                                const int* __restrict__ submap_inv_perm, double* __restrict__ submap_covs6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int pos = submap_inv_perm[offset + __float_as_int(kf_sorted[i].w)];
  const double* s = kf_covs6 + (size_t)i * 6;
  double* d = submap_covs6 + (size_t)pos * 6;
#pragma unroll
  for (int e = 0; e < 6; ++e) d[e] = s[e];
}

// pcl::transformPointCloud with a float 4x4 (column-major here).  PCL is not part of /root/reference (nor installed here), so the
// evaluation order is RECALLED from PCL >= 1.9's pcl/common/impl/transforms.hpp, not pinned: on x86-64 (SSE2, every DLO host)
// detail::Transformer<float>::se3 computes  x*c0 + (y*c1 + (z*c2 + c3))  with c_j the matrix columns; the scalar fallback
// computes ((x*m00 + y*m01) + z*m02) + m03.  The two differ by <= 1 ulp of the result.  This kernel follows the SSE2 order and
// is built with -ffp-contract=off (no FMA, like the reference's build: CMakeLists.txt:14-15).
__device__ __forceinline__ float3 transform_point_f(const float* __restrict__ m, float x, float y, float z) {
  float3 o;
  o.x = x * m[0] + (y * m[4] + (z * m[8] + m[12]));
  o.y = x * m[1] + (y * m[5] + (z * m[9] + m[13]));
  o.z = x * m[2] + (y * m[6] + (z * m[10] + m[14]));
  return o;
}

// cell-sorted device cloud -> transformed xyz in ORIGINAL point order (packed, 12 B per point)
__global__ void __launch_bounds__(256) k_transform_sorted_to_original(const float4* __restrict__ sorted, int n, const float* __restrict__ T_colmajor,
                                                                       float* __restrict__ out_xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = sorted[i];
  const float3 o = transform_point_f(T_colmajor, p.x, p.y, p.z);
  float* d = out_xyz + (size_t)__float_as_int(p.w) * 3;
  d[0] = o.x; d[1] = o.y; d[2] = o.z;
}

// cell-sorted device cloud -> transformed {x', y', z', 0} in ORIGINAL point order: the input of the submap voxel filter (odom.cc:1160-1163)
__global__ void __launch_bounds__(256) k_transform_sorted_to_original4(const float4* __restrict__ sorted, int n, const float* __restrict__ T_colmajor,
                                                                        float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = sorted[i];
  const float3 o = transform_point_f(T_colmajor, p.x, p.y, p.z);
  out[__float_as_int(p.w)] = make_float4(o.x, o.y, o.z, 0.f);
}

// cell-sorted device cloud -> transformed staging array {x', y', z', bitcast(original index)} (element = original index) + per-block
// bounding-box partials of the transformed points, ready for the index build: a keyframe made from a scan that is already on
// the device never visits the host (odom.cc:971-974 followed by :1166-1174).
__global__ void __launch_bounds__(256) k_transform_to_unsorted(const float4* __restrict__ sorted, int n, const float* __restrict__ T_colmajor,
                                                                float4* __restrict__ unsorted, float* __restrict__ bbox_part) {
  __shared__ float lds[4][6];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = sorted[i];
    const float3 o = transform_point_f(T_colmajor, p.x, p.y, p.z);
    unsorted[__float_as_int(p.w)] = make_float4(o.x, o.y, o.z, p.w);
    mn[0] = fminf(mn[0], o.x); mx[0] = fmaxf(mx[0], o.x);
    mn[1] = fminf(mn[1], o.y); mx[1] = fmaxf(mx[1], o.y);
    mn[2] = fminf(mn[2], o.z); mx[2] = fmaxf(mx[2], o.z);
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
  if (threadIdx.x == 6) bbox_part[blockIdx.x * 8 + 6] = 0.f;
}

// raw strided host layout (already on the device) -> transformed packed xyz, same point order
__global__ void __launch_bounds__(256) k_transform_raw(const unsigned char* __restrict__ raw, size_t stride_bytes, int n, const float* __restrict__ T_colmajor,
                                                        float* __restrict__ out_xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = reinterpret_cast<const float*>(raw + (size_t)i * stride_bytes);
  const float3 o = transform_point_f(T_colmajor, p[0], p[1], p[2]);
  float* d = out_xyz + (size_t)i * 3;
  d[0] = o.x; d[1] = o.y; d[2] = o.z;
}

// raw strided host layout (already on the device) -> float4 {x, y, z, intensity} (intensity 0 when the layout has none)
__global__ void __launch_bounds__(256) k_unpack_xyzi(const unsigned char* __restrict__ raw, size_t stride_bytes, long intensity_offset, int n, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* b = raw + (size_t)i * stride_bytes;
  const float* p = reinterpret_cast<const float*>(b);
  out[i] = make_float4(p[0], p[1], p[2], intensity_offset >= 0 ? *reinterpret_cast<const float*>(b + intensity_offset) : 0.f);
}

// filtered cloud {x, y, z, intensity} -> the index build's staging array {x, y, z, bitcast(index)} + per-block bounding-box partials
__global__ void __launch_bounds__(256) k_xyzi_to_unsorted(const float4* __restrict__ in, int n, float4* __restrict__ unsorted, float* __restrict__ bbox_part) {
  __shared__ float lds[4][6];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = in[i];
    unsorted[i] = make_float4(p.x, p.y, p.z, __int_as_float(i));
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

// Test hook: the small FP64 routines of ngicp_math.h evaluated ON THE DEVICE, one problem per thread (they are otherwise only
// reachable through whole alignments: so3_exp's Taylor branch, for one, only when a step happens to be < 1e-5 rad).
//   which 0: so3_exp_matrix   in 3  -> out 9      (gicp/so3.hpp:99-118 + Quaternion::toRotationMatrix)
//         1: ldlt6_solve      in 42 -> out 6      (A row-major 36, rhs 6; impl/lsq_registration_impl.hpp:147-148,172-173)
//         2: eig3_sym         in 6  -> out 12     (w 3, V 9; stands in for JacobiSVD, impl/nano_gicp_impl.hpp:332)
//         3: inv3_sym         in 6  -> out 6      (impl/nano_gicp_impl.hpp:205-209)
__global__ void __launch_bounds__(64) k_math_selftest(int which, const double* __restrict__ in, int n_problems, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_problems) return;
  if (which == 0) {
    double w[3], R[9];
    for (int e = 0; e < 3; ++e) w[e] = in[i * 3 + e];
    so3_exp_matrix(w, R);
    for (int e = 0; e < 9; ++e) out[i * 9 + e] = R[e];
  } else if (which == 1) {
    double A[36], rhs[6], x[6];
    for (int e = 0; e < 36; ++e) A[e] = in[i * 42 + e];
    for (int e = 0; e < 6; ++e) rhs[e] = in[i * 42 + 36 + e];
    ldlt6_solve(A, rhs, x);
    for (int e = 0; e < 6; ++e) out[i * 6 + e] = x[e];
  } else if (which == 2) {
    double C[6], w[3], V[9];
    for (int e = 0; e < 6; ++e) C[e] = in[i * 6 + e];
    eig3_sym(C, w, V);
    for (int e = 0; e < 3; ++e) out[i * 12 + e] = w[e];
    for (int e = 0; e < 9; ++e) out[i * 12 + 3 + e] = V[e];
  } else {
    double C[6], M[6];
    for (int e = 0; e < 6; ++e) C[e] = in[i * 6 + e];
    inv3_sym(C, M);
    for (int e = 0; e < 6; ++e) out[i * 6 + e] = M[e];
  }
}

// device stream copy (the measured copy bandwidth reported next to the nominal HBM peak): one block moves a
// contiguous 16 KiB piece, four float4 per thread in flight
__global__ void __launch_bounds__(256) k_stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = base + j * 256 < n16 ? src[base + j * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j * 256 < n16) dst[base + j * 256] = v[j];
}

}  // namespace ngk
