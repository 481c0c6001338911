// Which lanes of a wave64 LDS instruction share a conflict group, and over how many banks, for the two 64-bit operations
// the sweep kernels live on (ds_read_b64 gathers, ds_add_f64 scatters)? Sixteen waves of one workgroup per CU issue the same
// lane pattern (offset per wave), so the numbers are LDS throughput, not one wave's issue rate. Each pattern is built to
// separate the hypotheses "groups of 16 / 32 contiguous lanes" x "16 / 32 banks of 8 bytes".
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_group_probe.hip -o tools/build/lds_group_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int LDS_DOUBLES = 8192;   // 64 kB
constexpr int ITERS = 2048;

template <bool ATOMIC>
__global__ __launch_bounds__(1024) void probe(const int* __restrict__ lane_index, long long* cycles, double* sink) {
  __shared__ double lds[LDS_DOUBLES];
  for (int i = threadIdx.x; i < LDS_DOUBLES; i += 1024) lds[i] = 1.0;
  __syncthreads();
  const int idx = (lane_index[threadIdx.x & 63] + 512 * (threadIdx.x >> 6)) & (LDS_DOUBLES - 1);
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  const long long c0 = wall_clock64();
  for (int it = 0; it < ITERS; it += 8) {
#pragma unroll
    for (int x = 0; x < 8; x++) {
      if (ATOMIC) __hip_atomic_fetch_add(lds + idx, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else acc[x] += *(volatile double*)(lds + idx);
    }
  }
  __syncthreads();
  const long long c1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = c1 - c0;
  sink[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7] + lds[idx];
}

int main() {
  int* d_idx; long long* d_cyc; double* d_sink;
  hipMalloc(&d_idx, 64 * sizeof(int)); hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 256 * 1024 * 8);
  struct Pat { const char* name; std::vector<int> idx; };
  std::vector<Pat> pats;
  auto add = [&](const char* name, auto f) { Pat p; p.name = name; for (int l = 0; l < 64; l++) p.idx.push_back(f(l)); pats.push_back(p); };
  add("linear                                   ", [](int l) { return l; });
  add("l&15 + 32*(l>>4)  (16-groups share banks)", [](int l) { return (l & 15) + 32 * (l >> 4); });
  add("l&31 + 64*(l>>5)  (halves share banks)   ", [](int l) { return (l & 31) + 64 * (l >> 5); });
  add("l&7 + 16*(l>>3)   (l, l+8 same mod 16)   ", [](int l) { return (l & 7) + 16 * (l >> 3); });
  add("l&7 + 32*(l>>3)   (l, l+8 same mod 32)   ", [](int l) { return (l & 7) + 32 * (l >> 3); });
  add("stride 2                                 ", [](int l) { return 2 * l; });
  add("stride 4                                 ", [](int l) { return 4 * l; });
  add("l&3 + 32*(l>>2)   (4-way in 16, mod 32)  ", [](int l) { return (l & 3) + 32 * (l >> 2); });
  add("l&1 + 32*(l>>1)   (8-way in 16, mod 32)  ", [](int l) { return (l & 1) + 32 * (l >> 1); });
  add("32*l              (all one bank)         ", [](int l) { return 32 * l % 1024 + (32 * l / 1024); });
  add("pairs on one address (l>>1)              ", [](int l) { return l >> 1; });
  add("l, l+16 on one address: (l&15)+16*(l>>5) ", [](int l) { return (l & 15) + 16 * (l >> 5); });
  add("l, l+32 on one address: l&31             ", [](int l) { return l & 31; });
  for (unsigned seed : {12345u, 999u, 4242u}) {
    static char names[3][48]; static int n = 0;
    snprintf(names[n], 48, "random labels over 8192 (seed %5u)       ", seed);
    Pat p; p.name = names[n++]; unsigned x = seed;
    for (int l = 0; l < 64; l++) { x = x * 1664525u + 1013904223u; p.idx.push_back((x >> 8) % 1024); }
    pats.push_back(p);
  }
  // a random draw made conflict-free per 16-lane group over 16 banks / per 32-lane half over 32 banks (rows stay random)
  for (int nb : {16, 32}) {
    static char names[2][48]; static int n = 0;
    snprintf(names[n], 48, "random rows, banks distinct per %2d lanes  ", nb);
    Pat p; p.name = names[n++]; unsigned x = 777;
    for (int l = 0; l < 64; l++) { x = x * 1664525u + 1013904223u; const int row = (x >> 8) % (1024 / nb); p.idx.push_back(row * nb + (l % nb)); }
    pats.push_back(p);
  }
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount;
  for (auto& p : pats) {
    hipMemcpy(d_idx, p.idx.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    long long c[2];
    hipLaunchKernelGGL(probe<false>, dim3(grid), dim3(1024), 0, 0, d_idx, d_cyc, d_sink);
    hipMemcpy(&c[0], d_cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(probe<true>, dim3(grid), dim3(1024), 0, 0, d_idx, d_cyc, d_sink);
    hipMemcpy(&c[1], d_cyc, 8, hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz; 16 waves x ITERS instructions per CU
    const double ns_r = (double)c[0] * 10.0 / (16.0 * ITERS), ns_a = (double)c[1] * 10.0 / (16.0 * ITERS);
    printf("%s ds_read_b64 %6.2f ns  ds_add_f64 %6.2f ns per wave instruction (16 waves per CU, all CUs)\n", p.name, ns_r, ns_a);
  }
  return 0;
}
