// Timing probe for the dense Gauss-Jordan inverse (solver_kernels.hip) on a 1024 x 1024 diagonally dominant matrix:
// whole factorisation with HIP events, accuracy against the identity. GJ_PROBE=1: the look-ahead workgroup skips the
// inversion (timing only, wrong result); GJ_PROBE=2: no look-ahead work at all.
// Build (here, cross-compiled): hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DGJ_PROBE=n] tools/gj_probe.hip -Ikinetica_jl_amd/csrc -o tools/build/gj_probe[_n]
// Run on the GPU box: tools/build/gj_probe
#include "../kinetica_jl_amd/csrc/solver_kernels.hip"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 1024;
  std::vector<double> A((size_t)m * m);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) A[(size_t)i * m + j] = (i == j ? 8.0 : 0.0) + U(g) * (8.0 / m) * 4.0;
  double *S, *S0, *S2, *pinv; int* bad;
  hipMalloc(&S, sizeof(double) * m * m); hipMalloc(&S0, sizeof(double) * m * m); hipMalloc(&S2, sizeof(double) * m * m);
  hipMalloc(&pinv, sizeof(double) * 2 * 64 * 64); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipMemcpy(S0, A.data(), sizeof(double) * m * m, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double* out = nullptr;
  float best = 1e9f, sum = 0.f;
  const int reps = 20;
  for (int r = 0; r < reps + 3; r++) {
    hipMemcpyAsync(S, S0, sizeof(double) * m * m, hipMemcpyDeviceToDevice, s);
    hipEventRecord(e0, s);
    out = kin::launch_gauss_jordan(S, S2, m, pinv, bad, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
  }
  std::vector<double> Inv((size_t)m * m);
  hipMemcpy(Inv.data(), out, sizeof(double) * m * m, hipMemcpyDeviceToHost);
  int hbad; hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int i = 0; i < m; i += 7)
    for (int j = 0; j < m; j += 5) {
      double acc = 0.0;
      for (int k = 0; k < m; k++) acc += A[(size_t)i * m + k] * Inv[(size_t)k * m + j];
      worst = fmax(worst, fabs(acc - (i == j ? 1.0 : 0.0)));
    }
  unsigned long long fnv = 1469598103934665603ull;
  for (size_t i = 0; i < Inv.size(); i++) { unsigned long long w; memcpy(&w, &Inv[i], 8); fnv = (fnv ^ w) * 1099511628211ull; }
  printf("checksum of the inverse %016llx\n", fnv);
  printf("m=%d  GJ mean %.1f us  best %.1f us  (%d launches)  max|A inv - I| (sampled) %.2e  bad=%d\n", m, sum / reps * 1e3, best * 1e3,
         m / 32 + 1, worst, hbad);
  // ---- batched chains (round 4): n matrices per launch, checksum of every inverse against the single-matrix one
  for (int n : {1, 2, 4, 8, 16}) {
    std::vector<double*> Sb(n), S2b(n); std::vector<int*> badb(n);
    double* pb; hipMalloc(&pb, sizeof(double) * 2048 * n);
    for (int i = 0; i < n; i++) { hipMalloc(&Sb[i], sizeof(double) * m * m); hipMalloc(&S2b[i], sizeof(double) * m * m); badb[i] = bad; }
    float bsum = 0.f, bbest = 1e9f; int where = 0;
    for (int r = 0; r < 8; r++) {
      for (int i = 0; i < n; i++) hipMemcpyAsync(Sb[i], S0, sizeof(double) * m * m, hipMemcpyDeviceToDevice, s);
      hipEventRecord(e0, s);
      where = kin::launch_gauss_jordan_batched(n, Sb.data(), S2b.data(), m, pb, badb.data(), s);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r >= 3) { bsum += ms; bbest = ms < bbest ? ms : bbest; }
    }
    bool same = true;
    for (int i = 0; i < n; i++) {
      std::vector<double> Ib((size_t)m * m);
      hipMemcpy(Ib.data(), where ? S2b[i] : Sb[i], sizeof(double) * m * m, hipMemcpyDeviceToHost);
      same = same && memcmp(Ib.data(), Inv.data(), sizeof(double) * m * m) == 0;
    }
    printf("batched n=%2d: mean %.1f us  best %.1f us  = %.1f us per matrix, results %s\n", n, bsum / 5 * 1e3, bbest * 1e3, bbest * 1e3 / n,
           same ? "bit-identical to the single chain" : "DIFFER");
    for (int i = 0; i < n; i++) { hipFree(Sb[i]); hipFree(S2b[i]); }
    hipFree(pb);
  }
  return 0;
}
