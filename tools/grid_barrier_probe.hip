// Microbenchmark: cost of a software grid barrier (one atomic arrival counter in L2, bounded spin) for
// G co-resident workgroups on gfx950, against the cost of a dependent kernel boundary. Decides whether
// the solver's chains of tiny dependent kernels are worth fusing into persistent kernels.
// Build: hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 24)) { ok = false; break; }   // bounded: never hangs
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

__global__ void barrier_loop(unsigned* counter, int n, double* data, int* fail) {
  const unsigned G = gridDim.x;
  for (int i = 0; i < n; i++) {
    data[blockIdx.x * blockDim.x + threadIdx.x] += 1.0;   // a token amount of work between barriers
    if (!grid_barrier(counter, G * (unsigned)(i + 1))) { if (threadIdx.x == 0) *fail = 1; return; }
  }
}

__global__ void tiny(double* data) { data[blockIdx.x * blockDim.x + threadIdx.x] += 1.0; }

int main() {
  unsigned* counter; double* data; int* fail;
  hipMalloc(&counter, 4); hipMalloc(&data, 8 * 256 * 1024); hipMalloc(&fail, 4);
  hipMemset(data, 0, 8 * 256 * 1024); hipMemset(fail, 0, 4);
  hipStream_t s; hipStreamCreate(&s);
  for (int G : {32, 64, 128, 256}) {
    for (int T : {256, 1024}) {
      const int n = 200;
      double best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        hipMemsetAsync(counter, 0, 4, s);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(barrier_loop, dim3(G), dim3(T), 0, s, counter, n, data, fail);
        hipStreamSynchronize(s);
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      }
      int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
      printf("grid barrier  G=%3d x %4d threads: %.2f us per barrier (fail=%d)\n", G, T, best / n * 1e6, f);
    }
  }
  for (int G : {64, 256}) {
    const int n = 200;
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; i++) hipLaunchKernelGGL(tiny, dim3(G), dim3(256), 0, s, data);
      hipStreamSynchronize(s);
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("dependent kernel boundary G=%3d: %.2f us per kernel\n", G, best / n * 1e6);
  }
  return 0;
}
