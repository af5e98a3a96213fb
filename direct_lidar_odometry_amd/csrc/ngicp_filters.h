// Internal interface between the C-ABI translation unit (ngicp_api.hip) and the cloud filters (ngicp_filters.hip):
// removeNaN / CropBox / VoxelGrid on a device-resident float4 {x, y, z, intensity} cloud.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct FilterWorkspace {  // grow-only device buffers, owned by a handle (released by ngk_filter_free)
  void* buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int last_bits = 0;  // significant bits of the voxel indices of the previous call (how many sort passes the next one enqueues)
};

// in_dev: n points on the device.  remove_nan: drop non-finite points; crop_half > 0: drop the points inside the cube
// [-crop_half, crop_half]^3 (pcl::CropBox, negative); leaf > 0: pcl::VoxelGrid centroids.  *out_dev (device memory of the
// workspace, or in_dev itself when nothing was filtered) holds *n_out points.  Returns 0, or -1 with a message in err.
// One synchronisation of the stream, behind the last kernel (the counts that size each step stay on the device).
extern "C" int ngk_filter_cloud(hipStream_t stream, FilterWorkspace* ws, const float4* in_dev, int n, int remove_nan, float crop_half, float leaf,
                                const float4** out_dev, int* n_out, char* err, size_t errlen);
extern "C" void ngk_filter_free(FilterWorkspace* ws);
