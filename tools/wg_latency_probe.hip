// Latencies inside ONE 1024-thread workgroup on gfx950 (what bounds the phases of the resident integrator, resident.hip):
// barrier, dependent global load (L2 hit, small array re-read), LDS read, wave reduction by ds_bpermute / by DPP,
// noinline call. Build: hipcc -O3 --offload-arch=gfx950 tools/wg_latency_probe.hip -o tools/build/wg_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __noinline__ double callee(double x, int* sink) { if (threadIdx.x == 5000) *sink = 1; return x * 1.0000001 + 1e-9; }

__device__ __forceinline__ double wave_sum_bperm(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// reduction to lane 63 with DPP row shifts + row broadcasts (the classic GCN sequence)
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_mov<0x111>(v);   // row_shr:1
  v += dpp_mov<0x112>(v);   // row_shr:2
  v += dpp_mov<0x114>(v);   // row_shr:4
  v += dpp_mov<0x118>(v);   // row_shr:8
  v += dpp_mov<0x142>(v);   // row_bcast:15
  v += dpp_mov<0x143>(v);   // row_bcast:31
  return v;
}

__global__ __launch_bounds__(1024) void probe(const int* __restrict__ idx, double* data, long long* out, int iters, int n) {
  __shared__ double sh[2048];
  __shared__ int sink;
  const int tid = threadIdx.x;
  sh[tid] = tid; sh[tid + 1024] = 1.0;
  __syncthreads();
  long long t0, t1;
  // (a) barrier
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) __syncthreads();
  t1 = wall_clock64();
  if (tid == 0) out[0] = t1 - t0;
  // (b) dependent global loads: pointer chase through idx (n entries, L2 / L1 resident)
  int j = tid % n;
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) j = idx[j];
  t1 = wall_clock64();
  if (tid == 0) out[1] = t1 - t0;
  // (c) global load + barrier (a minimal phase): every thread reads one element, writes one, barrier
  double acc = 0.0;
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) { acc += data[(tid + i) % n]; data[n + tid] = acc; __syncthreads(); }
  t1 = wall_clock64();
  if (tid == 0) out[2] = t1 - t0;
  // (d) LDS dependent reads
  int q = tid;
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) q = (int)sh[q & 1023] ;
  t1 = wall_clock64();
  if (tid == 0) out[3] = t1 - t0;
  // (e) wave_sum by ds_bpermute, 5 values
  double v0 = acc + j + q, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4;
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) { v0 = wave_sum_bperm(v0) * 1e-3; v1 = wave_sum_bperm(v1) * 1e-3; v2 = wave_sum_bperm(v2) * 1e-3; v3 = wave_sum_bperm(v3) * 1e-3; v4 = wave_sum_bperm(v4) * 1e-3; }
  t1 = wall_clock64();
  if (tid == 0) out[4] = t1 - t0;
  // (f) the same by DPP
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) { v0 = wave_sum_dpp(v0) * 1e-3; v1 = wave_sum_dpp(v1) * 1e-3; v2 = wave_sum_dpp(v2) * 1e-3; v3 = wave_sum_dpp(v3) * 1e-3; v4 = wave_sum_dpp(v4) * 1e-3; }
  t1 = wall_clock64();
  if (tid == 0) out[5] = t1 - t0;
  // (g) noinline call
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) v0 = callee(v0, &sink);
  t1 = wall_clock64();
  if (tid == 0) out[6] = t1 - t0;
  // (h) global store + barrier + dependent global load of what another wave wrote (stage hand-over through L2)
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) { data[2 * n + tid] = v0 + i; __syncthreads(); v0 = data[2 * n + ((tid + 64) & 1023)]; __syncthreads(); }
  t1 = wall_clock64();
  if (tid == 0) out[7] = t1 - t0;
  // (i) the same through LDS
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) { sh[tid] = v0 + i; __syncthreads(); v0 = sh[(tid + 64) & 1023]; __syncthreads(); }
  t1 = wall_clock64();
  if (tid == 0) out[8] = t1 - t0;
  // (j) wall_clock64 itself
  long long s = 0;
  t0 = wall_clock64();
  for (int i = 0; i < iters; i++) s += wall_clock64();
  t1 = wall_clock64();
  if (tid == 0) { out[9] = t1 - t0; out[10] = s; }
  data[3 * n + tid] = v0 + v1 + v2 + v3 + v4 + j + q + acc;
}

int main() {
  const int n = 4096, iters = 2000;
  std::vector<int> h(n);
  for (int i = 0; i < n; i++) h[i] = (i * 1237 + 11) % n;
  int* d_idx; double* d_data; long long* d_out;
  CHECK(hipMalloc(&d_idx, n * sizeof(int))); CHECK(hipMalloc(&d_data, 8 * n * sizeof(double))); CHECK(hipMalloc(&d_out, 16 * sizeof(long long)));
  CHECK(hipMemcpy(d_idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
  CHECK(hipMemset(d_data, 0, 8 * n * sizeof(double)));
  const char* names[10] = {"barrier", "dependent global load (idx chase)", "global load + store + barrier", "dependent LDS read", "5 wave sums (ds_bpermute)",
                           "5 wave sums (DPP)", "noinline call", "stage hand-over through global memory (store, barrier, load, barrier)",
                           "stage hand-over through LDS", "wall_clock64 read"};
  for (int grid : {1, 256}) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, d_idx, d_data, d_out, iters, n);
      CHECK(hipDeviceSynchronize());
    }
    long long o[16];
    CHECK(hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost));
    printf("grid = %d workgroups of 1024 threads\n", grid);
    for (int i = 0; i < 10; i++) printf("  %-75s %8.1f ns\n", names[i], (double)o[i] * 10.0 / iters);
  }
  return 0;
}
