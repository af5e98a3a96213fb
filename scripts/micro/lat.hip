// Build and run (on the GPU box; the binary is not kept in the repository):  hipcc --offload-arch=gfx950 -O3 scripts/micro/lat.hip -o scripts/micro/lat && scripts/micro/lat
// Microbenchmark: dependent random 16-B gathers (pointer chase through a permutation) -> ns per access,
// for several table sizes and wave counts.  Also a single-lane dependent FP64 FMA chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void chase(const int4* __restrict__ tab, int n, int steps, int* out, int lanes_active) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if ((threadIdx.x & 63) >= lanes_active) return;
  unsigned idx = (unsigned)((tid * 2654435761u) % (unsigned)n);
  int acc = 0;
  for (int s = 0; s < steps; ++s) {
    int4 v = tab[idx];
    idx = (unsigned)v.x;
    acc += v.y;
  }
  out[tid] = acc + (int)idx;
}

__global__ void fma_chain(double* out, int steps) {
  double x = out[threadIdx.x], y = 1.0000001;
  if (threadIdx.x == 0) {
    for (int s = 0; s < steps; ++s) x = x * y + 1e-9;
    out[0] = x;
  }
}
__global__ void div_chain(double* out, int steps) {
  double x = out[threadIdx.x] + 3.0;
  if (threadIdx.x == 0) {
    for (int s = 0; s < steps; ++s) x = 1.0 / (x + 1.5);
    out[0] = x;
  }
}

int main() {
  const int steps = 64;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (size_t mb : {1, 8, 32, 128, 512}) {
    int n = (int)(mb * 1024 * 1024 / 16);
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_int4(perm[i], i, 0, 0);
    int4* d; int* o;
    CK(hipMalloc(&d, (size_t)n * 16));
    CK(hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    for (int blocks : {256, 1024, 4096}) {
      for (int lanes : {64, 16, 1}) {
        CK(hipMalloc(&o, (size_t)blocks * 256 * 4));
        chase<<<blocks, 256>>>(d, n, steps, o, lanes);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        chase<<<blocks, 256>>>(d, n, steps, o, lanes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("table %4zu MB blocks %5d (waves %6d) lanes/wave %2d: kernel %8.1f us -> %7.1f ns per dependent access, %6.2f G acc/s\n", mb, blocks, blocks * 4, lanes,
               ms * 1e3, ms * 1e6 / steps, (double)blocks * 4 * lanes * steps / (ms * 1e-3) / 1e9);
        CK(hipFree(o));
      }
    }
    CK(hipFree(d));
  }
  double* dd; CK(hipMalloc(&dd, 64 * 8)); CK(hipMemset(dd, 0, 64 * 8));
  for (int it = 0; it < 2; ++it) {
    CK(hipEventRecord(a)); fma_chain<<<1, 64>>>(dd, 10000); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("fp64 fma dependent chain: %.2f ns per op\n", ms * 1e6 / 10000);
    CK(hipEventRecord(a)); div_chain<<<1, 64>>>(dd, 10000); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("fp64 div dependent chain: %.2f ns per op\n", ms * 1e6 / 10000);
  }
  // empty kernel launch + sync latency
  for (int it = 0; it < 2; ++it) {
    CK(hipEventRecord(a));
    for (int k = 0; k < 100; ++k) fma_chain<<<1, 64>>>(dd, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("100 back-to-back tiny launches: %.2f us each\n", ms * 1e3 / 100);
  }
  return 0;
}
