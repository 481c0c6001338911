// Microbenchmark: cycles per wave64 LDS instruction (ds_read_b64, ds_add_f64) for strided and
// same-address lane patterns on gfx950 - how 64-bit accesses map to banks and what a same-address
// atomic costs. Background for the sweep kernels' LDS bound.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_bank_probe.hip -o /tmp/lbp && /tmp/lbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int LDS_DOUBLES = 8192;   // 64 kB
constexpr int ITERS = 4096;

template <bool ATOMIC>
__global__ __launch_bounds__(64) void probe(const int* __restrict__ lane_index, long long* cycles, double* sink) {
  __shared__ double lds[LDS_DOUBLES];
  for (int i = threadIdx.x; i < LDS_DOUBLES; i += 64) lds[i] = 1.0;
  __syncthreads();
  const int idx = lane_index[threadIdx.x];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it += 8) {
    // eight independent instructions per trip: throughput, not latency
#pragma unroll
    for (int x = 0; x < 8; x++) {
      if (ATOMIC) __hip_atomic_fetch_add(lds + idx, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else acc[x] += *(volatile double*)(lds + idx);
    }
  }
  __syncthreads();
  const long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) *cycles = c1 - c0;
  sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7] + lds[idx];
}

int main() {
  int* d_idx; long long* d_cyc; double* d_sink;
  hipMalloc(&d_idx, 64 * sizeof(int)); hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 64 * 8);
  struct Pat { const char* name; std::vector<int> idx; };
  std::vector<Pat> pats;
  for (int stride : {1, 2, 4, 8, 16, 32, 64, 128}) {
    Pat p; static char names[16][32]; static int n = 0;
    snprintf(names[n], 32, "stride %3d doubles", stride); p.name = names[n++];
    for (int l = 0; l < 64; l++) p.idx.push_back((l * stride) % LDS_DOUBLES);
    pats.push_back(p);
  }
  { Pat p; p.name = "same address"; p.idx.assign(64, 5); pats.push_back(p); }
  { Pat p; p.name = "8 lanes per address"; for (int l = 0; l < 64; l++) p.idx.push_back((l / 8) * 17); pats.push_back(p); }
  { Pat p; p.name = "random (lcg)"; unsigned x = 12345; for (int l = 0; l < 64; l++) { x = x * 1664525u + 1013904223u; p.idx.push_back((x >> 8) % LDS_DOUBLES); } pats.push_back(p); }
  { Pat p; p.name = "random (lcg 2)"; unsigned x = 999; for (int l = 0; l < 64; l++) { x = x * 1664525u + 1013904223u; p.idx.push_back((x >> 8) % LDS_DOUBLES); } pats.push_back(p); }
  for (auto& p : pats) {
    hipMemcpy(d_idx, p.idx.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    long long c[2];
    hipLaunchKernelGGL(probe<false>, dim3(1), dim3(64), 0, 0, d_idx, d_cyc, d_sink);
    hipMemcpy(&c[0], d_cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(probe<true>, dim3(1), dim3(64), 0, 0, d_idx, d_cyc, d_sink);
    hipMemcpy(&c[1], d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s ds_read_b64 %7.1f  ds_add_f64 %7.1f   cycles per wave instruction (one wave, 8 independent per trip)\n", p.name,
           (double)c[0] / ITERS, (double)c[1] / ITERS);
  }
  return 0;
}
