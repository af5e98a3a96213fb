// Build and run (on the GPU box; the binary is not kept in the repository):  hipcc --offload-arch=gfx950 -O3 scripts/micro/vmem_issue.hip -o scripts/micro/vmem_issue && scripts/micro/vmem_issue
// Microbenchmark: what does ONE scattered 16-byte gather wave-instruction cost a CU, by number of ACTIVE lanes and by how the
// 12 loads of a "window" are laid out (12 consecutive float4 per lane = the pass kernel's walk window)?
// Each wave issues `steps` windows; a window = 12 global_load_dwordx4 at consecutive addresses from a per-lane random base
// in an L2-resident table.  Reports cycles of CU time per wave-instruction (= kernel time * clock / instructions per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int W>
__global__ void __launch_bounds__(256) windows(const float4* __restrict__ tab, int n, int steps, float* out, int lanes_active) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (lane >= lanes_active) return;
  unsigned idx = (unsigned)((tid * 2654435761u) % (unsigned)(n - 64));
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    float4 c[W];
#pragma unroll
    for (int j = 0; j < W; ++j) c[j] = tab[idx + j];
    float m = 3e38f;
#pragma unroll
    for (int j = 0; j < W; ++j) m = fminf(m, c[j].x * c[j].x + c[j].y);
    acc += m;
    idx = (unsigned)((idx * 1664525u + 1013904223u + (unsigned)__float_as_int(m)) % (unsigned)(n - 64));  // dependent on the data
  }
  out[tid] = acc;
}

int main() {
  const int steps = 32;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (size_t mb : {1, 2, 4, 8, 32, 128}) {
  const int n = (int)(mb * 1024 * 1024 / 16);
  printf("---- table %zu MiB\n", mb);
  std::vector<float4> h(n);
  for (int i = 0; i < n; ++i) h[i] = make_float4((float)(i % 97) * 0.01f, (float)(i % 13), 0.f, 0.f);
  float4* d; float* o;
  CK(hipMalloc(&d, (size_t)n * 16));
  CK(hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
  const int blocks = 256 * 3;  // 3 blocks of 4 waves per CU = 12 waves per CU, one residency
  CK(hipMalloc(&o, (size_t)blocks * 256 * 4));
  for (int lanes : {64, 16, 1}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a));
      windows<12><<<blocks, 256>>>(d, n, steps, o, lanes);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
    }
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = 12.0 * steps * 12.0;  // waves per CU x steps x loads per window
    printf("active lanes %2d: %.1f us; %.1f ns per wave-instruction per CU (~%.0f cycles at 2.4 GHz); %.2f G lane-requests/s\n", lanes, ms * 1e3,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4, (double)blocks * 4 * lanes * steps * 12 / (ms * 1e-3) / 1e9);
    printf("    -> %.0f ns per dependent window step\n", ms * 1e6 / steps);
  }
  CK(hipFree(d)); CK(hipFree(o));
  }
  return 0;
}
