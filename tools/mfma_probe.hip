// Discovers the v_mfma_f64_16x16x4_f64 operand/accumulator layout on gfx950 with one-hot inputs.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(int* out) {   // block (la, lb): a one-hot at lane la, b one-hot at lane lb
  const int lane = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, c, 0, 0, 0);
  for (int v = 0; v < 4; v++) if (c[v] != 0.0) out[la * 64 + lb] = lane * 4 + v;
}
int main() {
  int* d; static int h[4096];
  hipMalloc(&d, sizeof h); hipMemset(d, 0xff, sizeof h);
  probe<<<dim3(64, 64), 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  // for each la: which lbs give a nonzero (same k), and where
  for (int la = 0; la < 64; la += 1) {
    printf("la=%2d: ", la);
    for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf("(lb=%d -> lane %d v %d) ", lb, h[la * 64 + lb] / 4, h[la * 64 + lb] % 4);
    printf("\n");
    if (la == 3) la = 14; if (la == 17) la = 30; if (la == 33) la = 46; if (la == 49) la = 61;
  }
  return 0;
}
