// gfx950 kernels of the implicit solve: Newton-matrix assembly, the dense Schur-block inverse
// (blocked Gauss-Jordan, FP64), GEMV, and the vector kernels of the variable-order BDF.
// Every kernel that belongs to a Newton iteration takes a device `skip` flag so that a whole
// iteration can be enqueued ahead of the convergence decision and become a no-op afterwards
// (one host synchronisation per step attempt instead of one per Newton iteration).
#include "lu.hpp"
#include "solver_kernels.hpp"
#include "ensemble.hpp"

#include "exp_tab.hpp"
#include "gj_dev.hpp"
#include "kernels.hpp"
#include "segsum_dev.hpp"

#include <utility>

namespace kin {

// ------------------------------------------------------------------------------------------
// M = I - c*J scattered into the factor storage
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lu_assemble_kernel(int nnzJ, const int32_t* __restrict__ jmap,
                                                          const double* __restrict__ jvals, double c,
                                                          double* __restrict__ W, long long off_S, int m, int mpad) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < nnzJ) {
    const int32_t jm = jmap[e];
    const bool diag = jm < 0;
    W[jm & 0x7fffffff] = (diag ? 1.0 : 0.0) - c * jvals[e];
  } else {
    const int d = m + (e - nnzJ);  // identity on the padding rows of the dense block
    if (d < mpad) W[off_S + (long long)d * mpad + d] = 1.0;
  }
}

void launch_lu_assemble(int64_t nnzJ, const int32_t* jmap, const double* jvals, double c, double* W,
                        int64_t off_S, int32_t m, int32_t mpad, hipStream_t s) {
  const int64_t total = nnzJ + (mpad - m);
  if (total == 0) return;
  hipLaunchKernelGGL(lu_assemble_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, (int)nnzJ, jmap, jvals,
                     c, W, (long long)off_S, m, mpad);
  KIN_HIP(hipGetLastError());
}

// L[:, p] /= diag[p] for the pivots of one round
// A pivot counts as vanished when a multiplier exceeds PIVOT_GROWTH_MAX in magnitude (or is not finite), or when the pivot
// itself is below PIVOT_MIN in magnitude: the matrix is I - c J, whose natural scale is 1 (its pivots stay >= ~1 unless an
// eigenvalue of c J comes close to 1, i.e. the matrix is close to singular).

__global__ __launch_bounds__(256) void lu_scale_kernel(int e0, int e1, const int32_t* __restrict__ ent_pivot, double* W,
                                                       long long off_L, long long off_diag, int* bad) {
  const int e = e0 + blockIdx.x * 256 + threadIdx.x;
  if (e >= e1) return;
  const double w = W[off_L + e], piv = W[off_diag + ent_pivot[e]];
  const double l = w / piv;
  if (bad && (!(fabs(piv) >= PIVOT_MIN) || (w != 0.0 && !(fabs(l) <= PIVOT_GROWTH_MAX)))) *bad = 1;   // benign race: every writer stores the same value
  W[off_L + e] = l;
}

void launch_lu_scale(int64_t e0, int64_t e1, const int32_t* ent_pivot, double* W, int64_t off_L, int64_t off_diag, int* bad, hipStream_t s) {
  if (e1 <= e0) return;
  hipLaunchKernelGGL(lu_scale_kernel, dim3((unsigned)ceil_div(e1 - e0, 256)), dim3(256), 0, s, (int)e0, (int)e1, ent_pivot,
                     W, (long long)off_L, (long long)off_diag, bad);
  KIN_HIP(hipGetLastError());
}

// jd[i] = J[i][i] (copy of the Jacobian's diagonal kept with a factorisation)
__global__ __launch_bounds__(256) void jac_diag_kernel(int N, const double* __restrict__ jv, const int32_t* __restrict__ j_diag,
                                                       double* __restrict__ jd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) jd[i] = jv[j_diag[i]];
}

// Drift of the Newton matrix's diagonal since slot s was made: out[s] = max_i max(q, 1/q) - 1 with
// q = (1 - c_s Jd_s[i]) / (1 - c_s Jd_now[i]) - how far diag(I - c_s J) of the slot is from what today's Jacobian gives at
// the same c. One 1024-thread workgroup per slot (blockIdx.x = slot).
__global__ __launch_bounds__(1024) void slot_drift_kernel(int N, const double* __restrict__ jv, const int32_t* __restrict__ j_diag,
                                                          SlotDriftArgs a, double* __restrict__ out) {
  __shared__ double sh[17];
  const double* jd = a.jd[blockIdx.x];
  const double c = a.c[blockIdx.x];
  double worst = 1.0;
  if (jd) {
    for (int i = threadIdx.x; i < N; i += 1024) {
      const double m_old = 1.0 - c * jd[i], m_new = 1.0 - c * jv[j_diag[i]];
      const double q = m_old / m_new;
      const double dev = (q > 0.0) ? fmax(q, 1.0 / q) : 1e300;   // a sign change or a NaN counts as unbounded drift
      worst = fmax(worst, dev == dev ? dev : 1e300);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_down(worst, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0) {
    double w = 1.0;
    for (int i = 0; i < 16; i++) w = fmax(w, sh[i]);
    out[blockIdx.x] = w - 1.0;
  }
}

// out[0] = max_i |J[i][i]| (one workgroup): the scale of the Jacobian behind the LU cache's absolute reuse rule (solver.cpp)
__global__ __launch_bounds__(1024) void jac_diag_absmax_kernel(int N, const double* __restrict__ jv, const int32_t* __restrict__ j_diag,
                                                               double* __restrict__ out) {
  __shared__ double sh[16];
  double worst = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) { const double v = fabs(jv[j_diag[i]]); worst = fmax(worst, v == v ? v : 1e300); }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_down(worst, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0) {
    double w = 0.0;
    for (int i = 0; i < 16; i++) w = fmax(w, sh[i]);
    out[0] = w;
  }
}
void launch_jac_diag_absmax(int N, const double* jv, const int32_t* j_diag, double* out, hipStream_t s) {
  hipLaunchKernelGGL(jac_diag_absmax_kernel, dim3(1), dim3(1024), 0, s, N, jv, j_diag, out);
  KIN_HIP(hipGetLastError());
}

void launch_jac_diag(int N, const double* jv, const int32_t* j_diag, double* jd, hipStream_t s) {
  hipLaunchKernelGGL(jac_diag_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, N, jv, j_diag, jd);
}
void launch_slot_drift(int N, int n_slots, const double* jv, const int32_t* j_diag, const SlotDriftArgs& a, double* out, hipStream_t s) {
  if (n_slots > 0) hipLaunchKernelGGL(slot_drift_kernel, dim3(n_slots), dim3(1024), 0, s, N, jv, j_diag, a, out);
}

// dinv[p] = 1 / diag[p] for the sparse pivots
__global__ __launch_bounds__(256) void lu_recip_kernel(int n, const double* __restrict__ diag, double* __restrict__ dinv) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dinv[i] = 1.0 / diag[i];
}

// Entries of the explicit inverses of the sparse triangular blocks: out[dst[e]] = sum over the entry's monomials of
// sign * product of factor values (paths through the elimination DAG; structure built once in SparseLU::analyze)
__global__ __launch_bounds__(256) void lu_mono_kernel(int n_ent, const int32_t* __restrict__ ent_ptr, const int32_t* __restrict__ mono_ptr,
                                                      const int32_t* __restrict__ fac, const float* __restrict__ sign,
                                                      const int32_t* __restrict__ dst, double* W) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_ent) return;
  double acc = 0.0;
  for (int32_t m = ent_ptr[e]; m < ent_ptr[e + 1]; m++) {
    double prod = (double)sign[m];
    for (int32_t f = mono_ptr[m]; f < mono_ptr[m + 1]; f++) prod *= W[fac[f]];
    acc += prod;
  }
  W[dst[e]] = acc;
}

void launch_lu_recip(int n, const double* diag, double* dinv, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(lu_recip_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, n, diag, dinv);
}
void launch_lu_mono(int n_ent, const int32_t* ent_ptr, const int32_t* mono_ptr, const int32_t* fac, const float* sign,
                    const int32_t* dst, double* W, hipStream_t s) {
  if (n_ent > 0) hipLaunchKernelGGL(lu_mono_kernel, dim3((unsigned)ceil_div(n_ent, 256)), dim3(256), 0, s, n_ent, ent_ptr, mono_ptr, fac, sign, dst, W);
}

// ------------------------------------------------------------------------------------------
// Blocked Gauss-Jordan inverse of the dense Schur block (no pivoting), NB = 32, 2 m^3 flops.
// Block step k reads the current matrix X and writes the next one Y (ping-pong, so no tile ever
// reads what another workgroup of the same launch writes):
//     P  = X[k,k]^-1                                  (gj_pivot_kernel, one small workgroup)
//     Y[k,:]  = P * X[k,:]          , Y[k,k] = P
//     Y[i,:]  = X[i,:] - X[i,k] * Y[k,:]   , Y[i,k] = -X[i,k] * P          (gj_update_kernel)
// The update runs on the matrix cores (v_mfma_f64_16x16x4_f64): each 64x64 tile first forms its
// 32x64 slice of the row panel (P times X[k, tile columns]) into LDS, then applies the rank-32
// update. Operand layout probed on gfx950 (tools/mfma_probe.hip): A[i][k] in lane i + 16k,
// B[k][j] in lane j + 16k, D[i][j] in lane 16*(i % 4) + j, register i / 4.
// ------------------------------------------------------------------------------------------
constexpr int GJ_NB = 32;
typedef double gj_d4 __attribute__((ext_vector_type(4)));

constexpr int GJ_XS = 4 * (GJ_NB / 2) * (GJ_NB / 2 + 1);     // scratch doubles of gj_invert_block_lds
// Runs in the workgroup's FIRST wavefront only (the others return at once): the two 16x16 inversions are single-wave
// work anyway, and the six 16x16x16 products in between are one v_mfma_f64_16x16x4 tile each (4 instructions, operands
// from LDS) - no workgroup barrier anywhere, the LDS executes a wave's operations in order. (Before: the products as
// 16-term scalar dot products by all 256 threads, ~200 LDS reads per thread and five barriers: 2.6 of the 5.3 us the
// inversion added to every launch, tools/gj_probe.hip.)
__device__ __forceinline__ void gj_invert_block_lds(double (*A)[GJ_NB + 1], double* __restrict__ XS, double* __restrict__ pinv, int* bad) {
  constexpr int H = GJ_NB / 2, LDX = H + 1;
  if (threadIdx.x >= 64) return;
  double* X0 = XS;                     // A11^-1
  double* X1 = X0 + H * LDX;           // T = A21 A11^-1
  double* X2 = X1 + H * LDX;           // U = A11^-1 A12
  double* X3 = X2 + H * LDX;           // S, then B22
  const int lane = threadIdx.x;
  const int r = lane >> 2, c0 = (lane & 3) * 4;     // layout of the 16x16 inversions: lane = (row, 4 columns)
  const int li = lane & 15, lk = lane >> 4;         // MFMA operand / result layout (see the head of this section)
  bool vanished = false;
  {
    double a[4];
#pragma unroll
    for (int x = 0; x < 4; x++) a[x] = A[r][c0 + x];
    vanished |= gj_inv16_wave(a, lane);
#pragma unroll
    for (int x = 0; x < 4; x++) X0[r * LDX + c0 + x] = a[x];
  }
  __builtin_amdgcn_wave_barrier();
  {
    gj_d4 tt = {0.0, 0.0, 0.0, 0.0}, uu = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int k = 4 * ks + lk;
      tt = __builtin_amdgcn_mfma_f64_16x16x4f64(A[H + li][k], X0[k * LDX + li], tt, 0, 0, 0);
      uu = __builtin_amdgcn_mfma_f64_16x16x4f64(X0[li * LDX + k], A[k][H + li], uu, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; v++) { X1[(4 * v + lk) * LDX + li] = tt[v]; X2[(4 * v + lk) * LDX + li] = uu[v]; }
  }
  __builtin_amdgcn_wave_barrier();
  {
    gj_d4 ss = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int k = 4 * ks + lk;
      ss = __builtin_amdgcn_mfma_f64_16x16x4f64(X1[li * LDX + k], A[k][H + li], ss, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; v++) X3[(4 * v + lk) * LDX + li] = A[H + 4 * v + lk][H + li] - ss[v];
  }
  __builtin_amdgcn_wave_barrier();
  {
    double a[4];
#pragma unroll
    for (int x = 0; x < 4; x++) a[x] = X3[r * LDX + c0 + x];
    vanished |= gj_inv16_wave(a, lane);
#pragma unroll
    for (int x = 0; x < 4; x++) X3[r * LDX + c0 + x] = a[x];
  }
  __builtin_amdgcn_wave_barrier();
  double* Y = &A[0][0];                // B12 (the input block is not needed any more)
  {
    gj_d4 p21 = {0.0, 0.0, 0.0, 0.0}, p12 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int k = 4 * ks + lk;
      p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(X3[li * LDX + k], X1[k * LDX + li], p21, 0, 0, 0);
      p12 = __builtin_amdgcn_mfma_f64_16x16x4f64(X2[li * LDX + k], X3[k * LDX + li], p12, 0, 0, 0);
    }
    double b22[4];
#pragma unroll
    for (int v = 0; v < 4; v++) b22[v] = X3[(4 * v + lk) * LDX + li];
    __builtin_amdgcn_wave_barrier();   // Y overlays nothing that is still read: A is dead since S was formed
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = 4 * v + lk;
      Y[i * LDX + li] = -p12[v];
      pinv[(H + i) * GJ_NB + li] = -p21[v];
      pinv[i * GJ_NB + H + li] = -p12[v];
      pinv[(H + i) * GJ_NB + H + li] = b22[v];
    }
  }
  __builtin_amdgcn_wave_barrier();
  {
    gj_d4 bb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int k = 4 * ks + lk;
      bb = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[li * LDX + k], X1[k * LDX + li], bb, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; v++) pinv[(4 * v + lk) * GJ_NB + li] = X0[(4 * v + lk) * LDX + li] - bb[v];
  }
  if (bad && vanished) *bad = 1;                // benign race: every writer stores the same value
}

__device__ __forceinline__ void gj_pivot_body(const double* __restrict__ X, int ld, int kb, double* __restrict__ pinv, int* bad) {
  __shared__ double A[GJ_NB][GJ_NB + 1];
  __shared__ double XS[GJ_XS];
  const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
#pragma unroll
  for (int j = 0; j < 4; j++) A[r][c0 + j] = X[(size_t)(kb * GJ_NB + r) * ld + kb * GJ_NB + c0 + j];
  __syncthreads();
  gj_invert_block_lds(A, XS, pinv, bad);
}
__global__ __launch_bounds__(256) void gj_pivot_kernel(const double* __restrict__ X, int ld, int kb, double* __restrict__ pinv, int* bad) {
  gj_pivot_body(X, ld, kb, pinv, bad);
}
// Several matrices of one size in one launch (the lockstep ensemble's members that factorise at the same time, ensemble.cpp):
// blockIdx.z = matrix; the arithmetic of each is exactly the single-matrix kernel's.
__global__ __launch_bounds__(256) void gj_pivot_batched_kernel(GjBatch B, int ld, int kb) {
  const int z = blockIdx.z;
  gj_pivot_body(B.X[z], ld, kb, B.pinv[z], B.bad[z]);
}

// grid = (mpad/64, 1 + mpad/64): block row 0 (dispatched first) holds ONE active workgroup (blockIdx.x == 0) that
// looks ahead: it recomputes only the next pivot block of Y and inverts it into pinv_next while the
// regular workgroups update their tiles, which takes the pivot inversion off the critical path.
__device__ __forceinline__ void gj_update_body(const double* __restrict__ X, double* __restrict__ Y, int ld, int kb,
                                               int nblk, const double* __restrict__ pinv, double* __restrict__ pinv_next,
                                               int* bad) {
  __shared__ double RP[GJ_NB][64 + 2];                 // row panel slice of this tile's columns
  __shared__ double A[GJ_NB][GJ_NB + 1];               // look-ahead workgroup only
  __shared__ double XS[GJ_XS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int kr0 = kb * GJ_NB, kr1 = kr0 + GJ_NB;
  if (blockIdx.y == 0) {
    const int kn = kb + 1;
    if (blockIdx.x != 0 || kn >= nblk) return;
#if defined(GJ_PROBE) && GJ_PROBE == 2   // tools/gj_probe.hip: timing without the look-ahead workgroup (wrong result)
    return;
#endif
    const int n0 = kn * GJ_NB;                           // next pivot block: rows/cols n0..n0+31 (never pivot rows/cols of step kb)
    const int rb = w >> 1, cb = w & 1;
    // every global operand of this workgroup is requested up front (one load latency instead of two: the
    // look-ahead is the critical path of the launch)
    double pa[8], xb[8], xa[8], xd[4];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      pa[ks] = pinv[(16 * rb + li) * GJ_NB + 4 * ks + lk];
      xb[ks] = X[(size_t)(kr0 + 4 * ks + lk) * ld + n0 + 16 * cb + li];
      xa[ks] = X[(size_t)(n0 + 16 * rb + li) * ld + kr0 + 4 * ks + lk];
    }
#pragma unroll
    for (int v = 0; v < 4; v++) xd[v] = X[(size_t)(n0 + 16 * rb + 4 * v + lk) * ld + n0 + 16 * cb + li];
    {  // RP[0:32][0:32] = P * X[k rows, n0 + cols]
      gj_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks], xb[ks], acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; v++) RP[16 * rb + 4 * v + lk][16 * cb + li] = acc[v];
    }
    __syncthreads();
    {  // A[0] = X[n0 rows, n0 cols] - X[n0 rows, k cols] * RP
      gj_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks], RP[4 * ks + lk][16 * cb + li], acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; v++) A[16 * rb + 4 * v + lk][16 * cb + li] = xd[v] - acc[v];
    }
    __syncthreads();
#if defined(GJ_PROBE) && GJ_PROBE == 1   // timing without the inversion of the next pivot block (wrong result)
    return;
#endif
    gj_invert_block_lds(A, XS, pinv_next, bad);
    return;
  }
  const int r0 = (blockIdx.y - 1) * 64 + 16 * w, c0 = blockIdx.x * 64;
  const bool pivot_rows = (r0 >= kr0 && r0 < kr1);
  // every global operand of the tile is requested up front - the pivot-column panel `a` and the old tile values `xo` do
  // not depend on the row panel, and behind the barrier they would be a second round of load latency
  double a[8], xo[4][4];
#pragma unroll
  for (int ks = 0; ks < 8; ks++) a[ks] = X[(size_t)(r0 + li) * ld + kr0 + 4 * ks + lk];
#pragma unroll
  for (int blk = 0; blk < 4; blk++)
#pragma unroll
    for (int v = 0; v < 4; v++) xo[blk][v] = X[(size_t)(r0 + 4 * v + lk) * ld + c0 + 16 * blk + li];
  // ---- row panel: wave w forms columns 16w..16w+15 (two 16-row blocks), RP = P * X[k rows, cols]
  {
    const int jc = c0 + 16 * w + li;
    const bool pivot_cols = (c0 + 16 * w >= kr0 && c0 + 16 * w < kr1);   // 16-column block inside the pivot block
    double bx[8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) bx[ks] = X[(size_t)(kr0 + 4 * ks + lk) * ld + jc];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
      gj_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const double ap = pinv[(16 * rb + li) * GJ_NB + 4 * ks + lk];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, bx[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int i = 16 * rb + 4 * v + lk;                                  // panel row
        RP[i][16 * w + li] = pivot_cols ? pinv[i * GJ_NB + (jc - kr0)] : acc[v];
      }
    }
  }
  __syncthreads();
  // ---- rank-32 update of this wave's 16 rows x 64 columns
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {
    gj_d4 acc = {0.0, 0.0, 0.0, 0.0};
    if (!pivot_rows) {
#pragma unroll
      for (int ks = 0; ks < 8; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], RP[4 * ks + lk][16 * blk + li], acc, 0, 0, 0);
    }
    const int j = c0 + 16 * blk + li;
    const bool pivot_col = (j >= kr0 && j < kr1);
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = r0 + 4 * v + lk;
      const size_t off = (size_t)i * ld + j;
      if (pivot_rows) Y[off] = RP[i - kr0][16 * blk + li];
      else if (pivot_col) Y[off] = -acc[v];
      else Y[off] = xo[blk][v] - acc[v];
    }
  }
}
__global__ __launch_bounds__(256) void gj_update_kernel(const double* __restrict__ X, double* __restrict__ Y, int ld, int kb,
                                                        int nblk, const double* __restrict__ pinv, double* __restrict__ pinv_next,
                                                        int* bad) {
  gj_update_body(X, Y, ld, kb, nblk, pinv, pinv_next, bad);
}
__global__ __launch_bounds__(256) void gj_update_batched_kernel(GjBatch B, int ld, int kb, int nblk) {
  const int z = blockIdx.z;
  gj_update_body(B.X[z], B.Y[z], ld, kb, nblk, B.pinv[z], B.pinv_next[z], B.bad[z]);
}

// returns the buffer that holds the inverse (S or S2)
double* launch_gauss_jordan(double* S, double* S2, int32_t mpad, double* pinv, int* bad, hipStream_t s) {
  const int nblk = mpad / GJ_NB;
  double* X = S;
  double* Y = S2;
  double* p_cur = pinv;                 // two 32x32 slots: current / next pivot-block inverse
  double* p_next = pinv + GJ_NB * GJ_NB;
  hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(256), 0, s, X, mpad, 0, p_cur, bad);
  for (int kb = 0; kb < nblk; kb++) {
    hipLaunchKernelGGL(gj_update_kernel, dim3(mpad / 64, mpad / 64 + 1), dim3(256), 0, s, X, Y, mpad, kb, nblk, p_cur, p_next, bad);
    std::swap(X, Y);
    std::swap(p_cur, p_next);
  }
  KIN_HIP(hipGetLastError());
  return X;
}

// n matrices (S[i], S2[i]; pivot scratch of 2 x 32 x 32 doubles each at pinv + 2048 i): one chain of launches for all of
// them. Returns 0 when the inverses end up in S[i], 1 when in S2[i] (the same for all: one size).
int launch_gauss_jordan_batched(int n, double* const* S, double* const* S2, int32_t mpad, double* pinv, int* const* bad, hipStream_t s) {
  const int nblk = mpad / GJ_NB;
  GjBatch B{};
  for (int i = 0; i < n; i++) {
    B.X[i] = S[i]; B.Y[i] = S2[i];
    B.pinv[i] = pinv + (size_t)i * 2 * GJ_NB * GJ_NB; B.pinv_next[i] = B.pinv[i] + GJ_NB * GJ_NB;
    B.bad[i] = bad[i];
  }
  hipLaunchKernelGGL(gj_pivot_batched_kernel, dim3(1, 1, n), dim3(256), 0, s, B, mpad, 0);
  for (int kb = 0; kb < nblk; kb++) {
    hipLaunchKernelGGL(gj_update_batched_kernel, dim3(mpad / 64, mpad / 64 + 1, n), dim3(256), 0, s, B, mpad, kb, nblk);
    for (int i = 0; i < n; i++) { std::swap(B.X[i], B.Y[i]); std::swap(B.pinv[i], B.pinv_next[i]); }
  }
  KIN_HIP(hipGetLastError());
  return nblk & 1;
}

// x = S[0:m, 0:m] * y : one wavefront per row
__global__ __launch_bounds__(256) void gemv_kernel(const double* __restrict__ S, int ld, int m, const double* __restrict__ y,
                                                   double* __restrict__ x, const int* skip) {
  const int sk = skip ? *skip : 0;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const double* a = S + (size_t)row * ld;
  // four independent partial sums per lane: all loads of a trip are in flight together (the row is read once,
  // the kernel is one dependent-latency chain per row otherwise); fixed order -> bitwise reproducible
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  int j = lane;
  if (j + 192 < m) {   // first trip issued before the flag is tested: an active launch does not wait for the flag alone
    const double a0 = a[j], a1 = a[j + 64], a2 = a[j + 128], a3 = a[j + 192];
    const double y0 = y[j], y1 = y[j + 64], y2 = y[j + 128], y3 = y[j + 192];
    if (sk) return;
    acc0 += a0 * y0; acc1 += a1 * y1; acc2 += a2 * y2; acc3 += a3 * y3;
    j += 256;
  } else if (sk) return;
  for (; j + 192 < m; j += 256) {
    const double a0 = a[j], a1 = a[j + 64], a2 = a[j + 128], a3 = a[j + 192];
    const double y0 = y[j], y1 = y[j + 64], y2 = y[j + 128], y3 = y[j + 192];
    acc0 += a0 * y0; acc1 += a1 * y1; acc2 += a2 * y2; acc3 += a3 * y3;
  }
  for (; j < m; j += 64) acc0 += a[j] * y[j];
  double acc = (acc0 + acc1) + (acc2 + acc3);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) x[row] = acc;
}

void launch_gemv(const double* S, int32_t ld, int32_t m, const double* y, double* x, const int* skip, hipStream_t s) {
  if (m == 0) return;
  hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)ceil_div(m, 4)), dim3(256), 0, s, S, ld, m, y, x, skip);
  KIN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// BDF vector kernels (algorithm: solver.cpp). D is the backward-difference array [MAXD][N].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
  // fixed-order reduction over a 1024-thread workgroup
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < 16; i++) t += sh[i];
    sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}

// the same shape for a maximum of non-negative values (NaN entries are ignored: the callers report them separately)
__device__ __forceinline__ double block_max_1024(double v, double* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; i++) t = fmax(t, sh[i]);
    sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}

__global__ __launch_bounds__(256) void bdf_predict_kernel(int N, int order, const double* __restrict__ D, BdfCoef cf,
                                                          double atol, double rtol, double* __restrict__ y,
                                                          double* __restrict__ psi, double* __restrict__ d,
                                                          double* __restrict__ scale, BdfCtrl* ctrl) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  // the predictor opens a corrector attempt: it also clears the attempt's control block (what a
  // separate one-thread launch used to do)
  if (i == 0) {
    ctrl->newton_done = 0; ctrl->converged = 0; ctrl->n_iter = 0; ctrl->nonfinite = 0; ctrl->any_negative = 0; ctrl->ticket = 0;
    ctrl->dy_norm_old = 0.0; ctrl->dy_norm = 0.0; ctrl->err_norm = 0.0; ctrl->err_m_norm = 0.0; ctrl->err_p_norm = 0.0;
  }
  if (i >= N) return;
  double yp = D[i], ps = 0.0;
  for (int j = 1; j <= order; j++) {
    const double dj = D[(size_t)j * N + i];
    yp += dj;
    ps += dj * cf.gamma[j];
  }
  y[i] = yp;
  psi[i] = ps / cf.alpha[order];
  d[i] = 0.0;
  scale[i] = atol + rtol * fabs(yp);
}

// Reductions over the state are spread over ceil(N / 1024) workgroups of 256 threads (four elements per
// thread, all loads in flight): each workgroup stores its partial sums, the last one to arrive (ticket in
// BdfCtrl) adds them in workgroup order - bitwise reproducible - and takes the decision. A single
// 1024-thread workgroup walking the whole state took 17 us at N = 10k, most of it load latency.
#ifndef KIN_RED_ELEMS
#define KIN_RED_ELEMS 1024
#endif
constexpr int RED_ELEMS = KIN_RED_ELEMS;   // elements per workgroup
constexpr int RED_PT = RED_ELEMS / 256;     // per thread

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// Hand-over of the per-workgroup partial sums to the workgroup that arrives last, without a cache-flushing fence
// (`__threadfence()` = buffer_wbl2 + buffer_inv, ~3.5 us on gfx950 - a third of these kernels' duration): the partials
// are stored write-through past L2 (relaxed agent-scope stores = `sc1`), the storing lane drains them (`s_waitcnt
// vmcnt(0)`) and then takes its ticket with an agent-scope atomic; the workgroup whose ticket is the last one reads
// the partials with `sc1` loads (sum_partials) after its atomic has returned. One storing lane per workgroup, 8-byte
// granules, one workgroup per CU: the form MI355X_MICROARCH.md lists as valid for inter-workgroup hand-offs.
__device__ __forceinline__ void store_partial(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// true in exactly one workgroup per launch: the one that arrives last, after every partial is visible.
// Call from all threads; thread 0 must be the one that stored the partials (store_partial).
__device__ __forceinline__ bool last_block_arrives(BdfCtrl* ctrl, int* flag) {
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(&ctrl->ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t == (int)gridDim.x - 1);
    if (*flag) __hip_atomic_store(&ctrl->ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
  }
  __syncthreads();
  return *flag != 0;
}

__device__ __forceinline__ double sum_partials(const double* part, int n) {
  double t = 0.0;
  for (int g = 0; g < n; g++) t += __hip_atomic_load(part + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past this CU's L1
  return t;
}

// A workgroup's five partial sums sit in ONE 128-byte line of their own (RED_SLOT doubles apart): write-through stores of 60
// workgroups into shared lines serialise at the memory side - 325 stores took ~45 us of a 55 us launch with the sums of all
// workgroups interleaved (part[q * G + g]), and ~7 of the 10.7 us of the 10-workgroup launch before it.
constexpr int RED_SLOT = 16;
// The five sums of a corrector launch from the workgroups' partial sums: by the first wavefront of the workgroup that arrived
// last, one partial per lane and round (all loads in flight together), fixed butterfly order - bitwise reproducible. (A
// single thread adding them one `sc1` load after the other was fine for 10 workgroups and is ~0.2-1 us per partial: the
// fused launch below has 60 workgroups at 10k species.)
__device__ __forceinline__ void newton_totals(const double* part, int G, double (&tot)[5]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 5; q++) {
    double v = 0.0;
    for (int g = lane; g < G; g += 64) v += __hip_atomic_load(part + g * RED_SLOT + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot[q] = wave_sum(v);
  }
}

// The decision of a corrector iteration, taken by ONE thread of the workgroup that arrived last (all partial sums are
// visible to it): norm of the update, contraction rate, converged / diverged / go on, and - when converged - the step's
// error-test norms; publishes the control block to the host when the attempt is decided (or `publish_always`).
struct NewtonDecide {
  int N, iter, maxit;
  double tol, rate_max, crate0, tol_first, dy_first_max;
  int crate_from_ctrl, ban_negatives;
  BdfCtrl* ctrl; const double* tot;   // tot[5]: the launch's sums (update, error test of order / -1 / +1, negative entries)
  BdfCtrl* host_ctrl; unsigned long long* host_seq; unsigned long long seq; int publish_always;
};
__device__ __forceinline__ void newton_decide(const NewtonDecide& a) {
  const int N = a.N, iter = a.iter, maxit = a.maxit, publish_always = a.publish_always;
  const double tol = a.tol, rate_max = a.rate_max, tol_first = a.tol_first, dy_first_max = a.dy_first_max;
  BdfCtrl* ctrl = a.ctrl; BdfCtrl* host_ctrl = a.host_ctrl;
  const double crate0 = (a.crate_from_ctrl && iter == 0) ? ctrl->crate : a.crate0;
  unsigned long long* host_seq = a.host_seq; const unsigned long long seq = a.seq;
  {
    const double tot = a.tot[0];
    const double old = ctrl->dy_norm_old;
    const double dy_norm = sqrt(tot / (double)N);
    const bool nonfinite = !isfinite(tot);
    const bool have_rate = iter > 0;
    const double rate = have_rate ? dy_norm / old : 0.0;
    // CVODE's carried convergence rate: every factorisation keeps the contraction it has shown (crate <- max(0.3 crate,
    // rate) after each iteration with a rate; 1 = unknown, set by the host when the factorisation is made). It lets the
    // FIRST iteration of a step be judged like the later ones instead of always being followed by a second one.
    double crate = iter == 0 ? crate0 : ctrl->crate;
    if (have_rate && !nonfinite) crate = fmax(0.3 * crate, rate);
    ctrl->crate = crate;
    bool diverged = nonfinite;
    // rate_max < 1 (a reused factorisation): a contraction slower than that means the matrix no longer matches the
    // Jacobian well enough for the error of the iteration to be judged from two or three corrections
    if (!diverged && have_rate) {
      double rp = rate;                                     // rate^(maxit - iter), 1 <= maxit - iter <= 3
      for (int e = 1; e < maxit - iter; e++) rp *= rate;
      if (rate >= rate_max || rp / (1.0 - rate) * dy_norm > tol) diverged = true;
    }
    ctrl->n_iter = iter + 1;
    ctrl->dy_norm = dy_norm;
    bool done = true, converged = false;
    if (diverged) { ctrl->nonfinite = nonfinite; }
    else if (dy_norm == 0.0 || (have_rate && rate / (1.0 - rate) * dy_norm < tol) ||
             (!have_rate && (dy_norm < tol || (crate0 < 1.0 && dy_norm <= dy_first_max && crate0 / (1.0 - crate0) * dy_norm < tol_first)))) {
      converged = true;                             // (first-iteration acceptance as in ode15s / CVODE)
    }
    else {
      ctrl->dy_norm_old = dy_norm;
      done = iter == maxit - 1;
    }
    if (converged) {
      const double te = a.tot[1];
      ctrl->err_norm = sqrt(te / (double)N);
      ctrl->err_m_norm = sqrt(a.tot[2] / (double)N);
      ctrl->err_p_norm = sqrt(a.tot[3] / (double)N);
      ctrl->any_negative = a.tot[4] > 0.0 ? (a.tot[4] >= BDF_NEG_MARK ? 3 : 1) : 0;   // bit 1: a species below -BDF_NEG_DEEP weights
      if (!isfinite(te)) ctrl->nonfinite = 1;
    }
    ctrl->converged = converged ? 1 : 0;
    ctrl->newton_done = done ? 1 : 0;
    // the verdict the host will reach from the same numbers (solver.cpp, step()): an accepted step with nothing that makes
    // the next one more than a continuation (every allowed iteration used = the host may drop the factorisation)
    ctrl->spec_go = (done && converged && !ctrl->nonfinite && !ctrl->lu_bad && !(a.ban_negatives && ctrl->any_negative) && !(ctrl->any_negative & 2) &&
                     !(ctrl->err_norm > 1.0) && iter + 1 < maxit) ? 1 : 0;
    if ((done || publish_always) && host_ctrl) {
      *host_ctrl = *ctrl;
      __threadfence_system();
      *(volatile unsigned long long*)host_seq = seq;
    }
  }
}

// One Newton update: y += dy, d += dy, ||dy||, convergence decision - and, folded in, the step's error estimate: every
// workgroup also reduces the error-test sums of the state THIS iteration produces, so the workgroup that takes the
// decision has them at hand when the decision is "converged" and publishes the attempt's result to the host itself
// (control block into coherent pinned host memory + sequence number the host spins on - no D2H copy, no stream
// synchronisation, and no separate error-estimate launch behind the corrector: 12 us per step at C3 size). The later
// launches of a blind-enqueued batch find newton_done set and return; `publish_always` marks the batch's last launch,
// which publishes "not decided yet" when that is the state. The update is applied unconditionally: when the decision is
// "diverged" the attempt is abandoned and y, d are rebuilt by the next predictor, so the stale update is never read.
__global__ __launch_bounds__(256) void bdf_newton_kernel(int N, int iter, int maxit, double tol, const int32_t* __restrict__ xloc,
                                                         const double* __restrict__ W, const double* __restrict__ scale,
                                                         double* __restrict__ y, double* __restrict__ d, double upd,
                                                         double rate_max, double crate0, double tol_first, double dy_first_max,
                                                         int order, const double* __restrict__ D, double atol, double rtol, BdfCoef cf,
                                                         BdfCtrl* ctrl, double* __restrict__ part, BdfCtrl* host_ctrl,
                                                         unsigned long long* host_seq, unsigned long long seq, int publish_always,
                                                         int crate_from_ctrl, int ban_negatives) {
  __shared__ double sh[20];
  __shared__ int last;
  // the "already decided" flag travels with the first round of loads instead of in front of it (a dependent load of its own is
  // ~1.5 us of every launch, segsum_dev.hpp): a launch behind the decision ends after ONE round of loads
  const int decided = __hip_atomic_load(&ctrl->newton_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int G = gridDim.x;
  // A non-finite update makes the sum of squares non-finite: one reduction carries both the norm and the flag.
  double s = 0.0, se = 0.0, sm = 0.0, sp = 0.0, neg = 0.0;
  const int i0 = blockIdx.x * RED_ELEMS + threadIdx.x;
  int32_t xl[RED_PT]; double dy[RED_PT], sc[RED_PT], yy[RED_PT], dd[RED_PT], dm[RED_PT], dp[RED_PT];
#pragma unroll
  for (int x = 0; x < RED_PT; x++) {
    const int i = i0 + 256 * x;
    const bool ok = i < N;
    xl[x] = ok ? xloc[i] : -1; sc[x] = ok ? scale[i] : 1.0;
    yy[x] = ok ? y[i] : 0.0; dd[x] = ok ? d[i] : 0.0;
    dm[x] = (ok && order > 1) ? D[(size_t)order * N + i] : 0.0;
    dp[x] = (ok && order < 5) ? D[(size_t)(order + 1) * N + i] : 0.0;
  }
  if (decided) return;
#pragma unroll
  for (int x = 0; x < RED_PT; x++) dy[x] = xl[x] >= 0 ? upd * W[xl[x]] : 0.0;   // upd = 2 / (1 + c / c_fact): reused factorisation
#pragma unroll
  for (int x = 0; x < RED_PT; x++) {
    const double q = dy[x] / sc[x];
    s += q * q;
    yy[x] += dy[x]; dd[x] += dy[x];
    if (i0 + 256 * x >= N) continue;
    // error test of the state after this iteration (a non-finite state makes the sum non-finite)
    const double sce = atol + rtol * fabs(yy[x]);
    if (yy[x] < 0.0) neg = fmax(neg, yy[x] < -BDF_NEG_DEEP * sce ? BDF_NEG_MARK : 1.0);
    const double e = cf.error_const[order] * dd[x] / sce;
    se += e * e + (isfinite(yy[x]) ? 0.0 : INFINITY);
    if (order > 1) { const double em = cf.error_const[order - 1] * (dm[x] + dd[x]) / sce; sm += em * em; }
    if (order < 5) { const double ep = cf.error_const[order + 1] * (dd[x] - dp[x]) / sce; sp += ep * ep; }
  }
  // five sums, one pair of barriers
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s += __shfl_down(s, off, 64); se += __shfl_down(se, off, 64); sm += __shfl_down(sm, off, 64);
    sp += __shfl_down(sp, off, 64); neg += __shfl_down(neg, off, 64);
  }
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[5 * w] = s; sh[5 * w + 1] = se; sh[5 * w + 2] = sm; sh[5 * w + 3] = sp; sh[5 * w + 4] = neg; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; q++) store_partial(part + blockIdx.x * RED_SLOT + q, (sh[q] + sh[5 + q]) + (sh[10 + q] + sh[15 + q]));
  }
  const bool is_last = last_block_arrives(ctrl, &last);
  // the state update is off the critical path of the decision: its stores go out while the ticket travels
#pragma unroll
  for (int x = 0; x < RED_PT; x++) {
    const int i = i0 + 256 * x;
    if (i < N) { y[i] = yy[x]; d[i] = dd[x]; }
  }
  if (!is_last || threadIdx.x >= 64) return;
  double tot[5];
  newton_totals(part, G, tot);
  if (threadIdx.x == 0)
    newton_decide(NewtonDecide{N, iter, maxit, tol, rate_max, crate0, tol_first, dy_first_max, crate_from_ctrl, ban_negatives, ctrl, tot,
                               host_ctrl, host_seq, seq, publish_always});
}

// ------------------------------------------------------------------------------------------
// The last gather stage of the fused solve (lu.hpp: x1 = V y1 + NVU x2) and the corrector update in ONE launch: a row of the
// stage produces the solution component of one species, and everything the update needs of that species (scale, y, d, two
// difference rows) is requested together with the row's descriptors, before the gather. The dense block's components
// (x2, already there from the GEMV) are taken by extra wavefront tasks behind the plan's. Partial sums, ticket and decision
// as in bdf_newton_kernel. One launch of ~11 us and its dependency gap less per corrector iteration (C3: 5.4 + 10.7 us
// -> see DESIGN 3.3). The plan's `aux` entries hold the species index of each row (lu.cpp: build_C).
// ------------------------------------------------------------------------------------------
struct NewtonElem { double sc, yy, dd, dm, dp; };
struct NewtonSums { double s = 0.0, se = 0.0, sm = 0.0, sp = 0.0, neg = 0.0; };

__device__ __forceinline__ NewtonElem newton_pre(const NewtonFuse& f, int32_t sp) {
  NewtonElem e{1.0, 0.0, 0.0, 0.0, 0.0};
  if (sp < 0) return e;
  e.sc = f.scale[sp]; e.yy = f.y[sp]; e.dd = f.d[sp];
  if (f.order > 1) e.dm = f.D[(size_t)f.order * f.N + sp];
  if (f.order < 5) e.dp = f.D[(size_t)(f.order + 1) * f.N + sp];
  return e;
}
// the update of species sp with solution component x (the arithmetic of bdf_newton_kernel, element by element)
__device__ __forceinline__ void newton_apply(const NewtonFuse& f, int32_t sp, double x, NewtonElem e, NewtonSums& t) {
  const double dy = f.upd * x;
  const double q = dy / e.sc;
  t.s += q * q;
  e.yy += dy; e.dd += dy;
  const double sce = f.atol + f.rtol * fabs(e.yy);
  if (e.yy < 0.0) t.neg = fmax(t.neg, e.yy < -BDF_NEG_DEEP * sce ? BDF_NEG_MARK : 1.0);
  const double er = f.ec * e.dd / sce;
  t.se += er * er + (isfinite(e.yy) ? 0.0 : INFINITY);
  if (f.order > 1) { const double em = f.ec_m * (e.dm + e.dd) / sce; t.sm += em * em; }
  if (f.order < 5) { const double ep = f.ec_p * (e.dd - e.dp) / sce; t.sp += ep * ep; }
  f.y[sp] = e.yy; f.d[sp] = e.dd;
}

template <int SEG_WG>
__global__ __launch_bounds__(SEG_WG) void stagec_newton_kernel(SegPlanView p, double* W, NewtonFuse f) {
  constexpr int OP = SEG_PROD_SET;
  constexpr int SEG_WAVES = SEG_WG / 64, BLK_PER_THREAD = SegPlanHost::BLK_PASS / 1024;
  __shared__ double sh[5 * SEG_WAVES];
  __shared__ double shb[SEG_WAVES];
  __shared__ int last;
  const int skip = *f.skip;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool impl = p.val_base >= 0;
  const SegExtra ex;
  NewtonSums t;
  if (SEG_WG == 1024 && (int)blockIdx.x < p.B) {       // one long row for the whole workgroup
    const int r = blockIdx.x;
    const int32_t e0 = p.blk_beg[r], e1 = p.blk_end[r];
    const int32_t bdst = p.blk_dst[r];
    const int32_t bsp = p.blk_aux[r];
    if (skip) return;      // (whole grid: the flag is uniform and only this launch's LAST workgroup can raise it)
    const NewtonElem pre = newton_pre(f, threadIdx.x == 0 ? bsp : -1);
    double acc = 0.0;
    for (int32_t base = e0; base < e1; base += SegPlanHost::BLK_PASS)
      acc += seg_gather<OP, BLK_PER_THREAD, false>(p, W, ex, impl, [&](int x) {
        const int32_t e = base + (int32_t)threadIdx.x + 1024 * x;
        return e < e1 ? e : -1;
      });
    acc = wave_sum(acc);
    if (lane == 0) shb[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < SEG_WAVES; w++) tot += shb[w];     // fixed order
      W[bdst] = tot;
      newton_apply(f, bsp, tot, pre, t);
    }
  } else {
    const int task = ((int)blockIdx.x - p.B) * SEG_WAVES + wv;
    if (task < p.G) {                                   // an ELL group: one short row per lane
      const int32_t dst = p.grp_dst[task * 64 + lane];
      const int32_t sp = dst >= 0 ? p.grp_aux[task * 64 + lane] : -1;
      const int32_t c0 = p.grp_off[task], c1 = p.grp_off[task + 1];
      if (skip) return;
      const NewtonElem pre = newton_pre(f, sp);
      double acc = 0.0;
      for (int32_t col = c0; col < c1; col += 8)
        acc += seg_gather<OP, 8, true>(p, W, ex, impl, [&](int x) { return col + x < c1 ? (col + x) * 64 + lane : -1; });
      if (dst >= 0) { W[dst] = acc; newton_apply(f, sp, acc, pre, t); }
    } else if (task < p.G + p.S) {                      // one medium row per wavefront
      const int sidx = task - p.G;
      const int32_t e0 = p.seg_beg[sidx], e1 = p.seg_end[sidx];
      const int32_t sdst = p.seg_dst[sidx];
      const int32_t ssp = p.seg_aux[sidx];
      if (skip) return;
      const NewtonElem pre = newton_pre(f, lane == 0 ? ssp : -1);
      double acc;
      if (e1 - e0 <= 256)
        acc = seg_gather<OP, 4, false>(p, W, ex, impl, [&](int x) { const int32_t e = e0 + lane + 64 * x; return e < e1 ? e : -1; });
      else
        acc = seg_gather<OP, 16, false>(p, W, ex, impl, [&](int x) { const int32_t e = e0 + lane + 64 * x; return e < e1 ? e : -1; });
      acc = wave_sum(acc);
      if (lane == 0) { W[sdst] = acc; newton_apply(f, ssp, acc, pre, t); }
    } else {                                            // the dense block's species: x2 is in place since the GEMV
      if (skip) return;
      const int j = (task - p.G - p.S) * 64 + lane;
      if (j < f.m) {
        const int32_t sp = f.x2_species[j];
        const NewtonElem pre = newton_pre(f, sp);
        newton_apply(f, sp, W[f.off_x + j], pre, t);
      }
    }
  }
  // five sums over the workgroup, one pair of barriers
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    t.s += __shfl_down(t.s, off, 64); t.se += __shfl_down(t.se, off, 64); t.sm += __shfl_down(t.sm, off, 64);
    t.sp += __shfl_down(t.sp, off, 64); t.neg += __shfl_down(t.neg, off, 64);
  }
  if (lane == 0) { sh[5 * wv] = t.s; sh[5 * wv + 1] = t.se; sh[5 * wv + 2] = t.sm; sh[5 * wv + 3] = t.sp; sh[5 * wv + 4] = t.neg; }
  __syncthreads();
  const int G = gridDim.x;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; q++) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < SEG_WAVES; w++) tot += sh[5 * w + q];     // fixed order
      store_partial(f.part + blockIdx.x * RED_SLOT + q, tot);
    }
  }
  if (!last_block_arrives(f.ctrl, &last) || threadIdx.x >= 64) return;
  double tot[5];
  newton_totals(f.part, G, tot);
  if (threadIdx.x == 0)
    newton_decide(NewtonDecide{f.N, f.iter, f.maxit, f.tol, f.rate_max, f.crate0, f.tol_first, f.dy_first_max, f.crate_from_ctrl,
                               f.ban_negatives, f.ctrl, tot, f.host_ctrl, f.host_seq, f.seq, f.publish_always});
}

static bool stagec_big_wg(const SegPlanView& p) {
  return p.B > 0;
}
int stagec_newton_grid(const SegPlanView& p, int m) {
  const int tasks = p.G + p.S + (int)ceil_div(m, 64);
  return stagec_big_wg(p) ? p.B + (int)ceil_div(tasks, 16) : (int)ceil_div(tasks, 4);
}
void launch_stagec_newton(const SegPlanView& p, double* W, const NewtonFuse& f, hipStream_t s) {
  const int grid = stagec_newton_grid(p, f.m);
  if (stagec_big_wg(p)) hipLaunchKernelGGL((stagec_newton_kernel<1024>), dim3(grid), dim3(1024), 0, s, p, W, f);
  else hipLaunchKernelGGL((stagec_newton_kernel<256>), dim3(grid), dim3(256), 0, s, p, W, f);
  KIN_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void bdf_accept_kernel(int N, int order, double* __restrict__ D, const double* __restrict__ d,
                                                         double* __restrict__ copy_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double di = d[i];
  D[(size_t)(order + 2) * N + i] = di - D[(size_t)(order + 1) * N + i];
  D[(size_t)(order + 1) * N + i] = di;
  double carry = di;
  for (int j = order; j >= 0; j--) {
    carry += D[(size_t)j * N + i];
    D[(size_t)j * N + i] = carry;
  }
  if (copy_out) copy_out[i] = carry;     // the new state D[0]
}

// The accept of the previous step and the predictor of the next one in one pass over the columns of D (the host defers
// the accept until it knows what follows it; when a step-size change or a dense-output save comes first, the plain
// accept kernel runs instead). `ao` = order of the accepted step, `order` = order of the step being predicted.
__global__ __launch_bounds__(256) void bdf_accept_predict_kernel(int N, int ao, int order, double* __restrict__ D, BdfCoef cf,
                                                                 double atol, double rtol, double* __restrict__ y,
                                                                 double* __restrict__ psi, double* __restrict__ d,
                                                                 double* __restrict__ scale, BdfCtrl* ctrl,
                                                                 double* __restrict__ copy_out, const int* go) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  // A speculatively enqueued step (go = &ctrl->spec_go): when the step before it did not end the way the host assumed, this
  // launch leaves everything alone and makes sure the iterations behind it stay no-ops (newton_done is 1 already unless the
  // batch before ended undecided; nothing in this launch writes spec_go, so every workgroup reads the same value).
  if (go && !*go) {
    if (i == 0) ctrl->newton_done = 1;
    return;
  }
  if (i == 0) {
    ctrl->newton_done = 0; ctrl->converged = 0; ctrl->n_iter = 0; ctrl->nonfinite = 0; ctrl->any_negative = 0; ctrl->ticket = 0;
    ctrl->dy_norm_old = 0.0; ctrl->dy_norm = 0.0; ctrl->err_norm = 0.0; ctrl->err_m_norm = 0.0; ctrl->err_p_norm = 0.0;
  }
  if (i >= N) return;
  double col[BDF_D_ROWS];
  const double di = d[i];
#pragma unroll
  for (int j = 0; j < BDF_D_ROWS; j++) col[j] = j <= ao + 1 ? D[(size_t)j * N + i] : 0.0;
  // accept (bdf_accept_kernel)
#pragma unroll
  for (int j = BDF_D_ROWS - 1; j >= 1; j--) {
    if (j == ao + 2) col[j] = di - col[j - 1];
  }
#pragma unroll
  for (int j = 0; j < BDF_D_ROWS; j++) if (j == ao + 1) col[j] = di;
  double carry = di;
#pragma unroll
  for (int j = BDF_D_ROWS - 1; j >= 0; j--) {
    if (j <= ao) { carry += col[j]; col[j] = carry; }
  }
#pragma unroll
  for (int j = 0; j < BDF_D_ROWS; j++) if (j <= ao + 2) D[(size_t)j * N + i] = col[j];
  if (copy_out) copy_out[i] = col[0];
  // predict (bdf_predict_kernel), order <= ao + 1
  double yp = col[0], ps = 0.0;
#pragma unroll
  for (int j = 1; j < BDF_D_ROWS; j++) {
    if (j <= order) { yp += col[j]; ps += col[j] * cf.gamma[j]; }
  }
  y[i] = yp;
  psi[i] = ps / cf.alpha[order];
  d[i] = 0.0;
  scale[i] = atol + rtol * fabs(yp);
}

// D[0..order] <- (R U)^T D[0..order]   (step-size change by `factor`, matrix built on the host)
__global__ __launch_bounds__(256) void bdf_change_D_kernel(int N, int order, BdfMat ru, double* __restrict__ D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double v[6], o[6];
  for (int j = 0; j <= order; j++) v[j] = D[(size_t)j * N + i];
  for (int a = 0; a <= order; a++) {
    double t = 0.0;
    for (int b = 0; b <= order; b++) t += ru.v[b][a] * v[b];
    o[a] = t;
  }
  for (int j = 0; j <= order; j++) D[(size_t)j * N + i] = o[j];
}

__global__ __launch_bounds__(256) void bdf_init_D_kernel(int N, int nrows, const double* __restrict__ y0, const double* __restrict__ f0,
                                                         double h, double* __restrict__ D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  D[i] = y0[i];
  D[(size_t)N + i] = f0[i] * h;
  for (int j = 2; j < nrows; j++) D[(size_t)j * N + i] = 0.0;
}

// dense output: out = D[0] + sum_j p[j] D[j]
__global__ __launch_bounds__(256) void bdf_interp_kernel(int N, int order, const double* __restrict__ D, BdfVec p, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double v = D[i];
  for (int j = 1; j <= order; j++) v += p.v[j] * D[(size_t)j * N + i];
  out[i] = v;
}

// ------------------------------------------------------------------------------------------
// Explicit Dormand-Prince 5(4) (C2: "RHS kernel only, explicit solver"): stage combination, error norm,
// dense output. K is the stage array [7][N]; algorithm and step control: solver.cpp (rk_step).
// ------------------------------------------------------------------------------------------
// out = y + sum_j w[j] K[j], j < n  (stage argument, new state, dense output: the weights come from the host)
__global__ __launch_bounds__(256) void rk_combine_kernel(int N, int n, RkVec w, const double* __restrict__ y,
                                                         const double* __restrict__ K, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double v = 0.0;
  for (int j = 0; j < n; j++) v += w.v[j] * K[(size_t)j * N + i];
  out[i] = y[i] + v;
}

// err_norm = rms( (sum_j e[j] K[j]) / (atol + rtol max(|y|, |y_new|)) ), flags for NaN / negative entries
__global__ __launch_bounds__(256) void rk_error_kernel(int N, RkVec e, const double* __restrict__ y, const double* __restrict__ y_new,
                                                       const double* __restrict__ K, double atol, double rtol, BdfCtrl* ctrl,
                                                       double* __restrict__ part, BdfCtrl* host_ctrl, unsigned long long* host_seq,
                                                       unsigned long long seq) {
  __shared__ double sh[4];
  __shared__ int last;
  const int G = gridDim.x;
  double se = 0.0;
  int neg = 0, bad = 0;
  for (int x = 0; x < RED_PT; x++) {
    const int i = blockIdx.x * RED_ELEMS + threadIdx.x + 256 * x;
    if (i >= N) continue;
    double err = 0.0;
    for (int j = 0; j < 7; j++) err += e.v[j] * K[(size_t)j * N + i];
    const double yn = y_new[i];
    if (yn < 0.0) neg = 1;
    if (!isfinite(yn)) bad = 1;
    const double q = err / (atol + rtol * fmax(fabs(y[i]), fabs(yn)));
    se += q * q;
  }
  const double pe = block_sum_256(se, sh), pn = block_sum_256((double)neg, sh), pb = block_sum_256((double)bad, sh);
  if (threadIdx.x == 0) { store_partial(part + blockIdx.x, pe); store_partial(part + G + blockIdx.x, pn); store_partial(part + 2 * G + blockIdx.x, pb); }
  if (!last_block_arrives(ctrl, &last)) return;
  if (threadIdx.x == 0) {
    ctrl->err_norm = sqrt(sum_partials(part, G) / (double)N);
    ctrl->any_negative = sum_partials(part + G, G) > 0.0;
    ctrl->nonfinite = sum_partials(part + 2 * G, G) > 0.0 || !isfinite(ctrl->err_norm);
    if (host_ctrl) {   // step-end hand-over through pinned host memory (see bdf_newton_kernel)
      *host_ctrl = *ctrl;
      __threadfence_system();
      *(volatile unsigned long long*)host_seq = seq;
    }
  }
}

// out = a + s*b
// out[i] = max(a[i], 0): the start state of a RETRIED chunk (see solve_entry)
__global__ __launch_bounds__(256) void clip_negative_kernel(int N, const double* __restrict__ a, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) { const double v = a[i]; out[i] = v < 0.0 ? 0.0 : v; }
}

__global__ __launch_bounds__(256) void axpy_out_kernel(int N, const double* __restrict__ a, double s, const double* __restrict__ b, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) out[i] = a[i] + s * b[i];
}

// norms for the initial step size: rms(y0/sc), rms(f0/sc), rms((f1-f0)/sc), max |f0| / (0.1 |y0| + sc), sc = atol + rtol |y0|;
// f1 may be null. Also reports non-finite f.
__global__ __launch_bounds__(1024) void bdf_norms_kernel(int N, const double* __restrict__ y0, const double* __restrict__ f0,
                                                         const double* __restrict__ f1, double atol, double rtol, BdfCtrl* ctrl) {
  __shared__ double sh[17];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, vm = 0.0;
  int bad = 0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double sc = atol + rtol * fabs(y0[i]);
    const double a = y0[i] / sc, b = f0[i] / sc;
    s0 += a * a; s1 += b * b;
    vm = fmax(vm, fabs(f0[i]) / (0.1 * fabs(y0[i]) + sc));
    if (!isfinite(f0[i])) bad = 1;
    if (f1) { const double c = (f1[i] - f0[i]) / sc; s2 += c * c; if (!isfinite(f1[i])) bad = 1; }
  }
  const double t0 = block_sum_1024(s0, sh), t1 = block_sum_1024(s1, sh), t2 = block_sum_1024(s2, sh);
  const double tb = block_sum_1024((double)bad, sh);
  const double tm = block_max_1024(vm, sh);
  if (threadIdx.x == 0) {
    ctrl->scratch[0] = sqrt(t0 / (double)N);
    ctrl->scratch[1] = sqrt(t1 / (double)N);
    ctrl->scratch[2] = sqrt(t2 / (double)N);
    ctrl->scratch[3] = tm;
    ctrl->nonfinite = tb > 0.0;
  }
}

// per-species running maximum over saved states (identify_next_seeds' reduction)
__global__ __launch_bounds__(256) void colmax_kernel(int N, long long M, const double* __restrict__ U, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double v = U[i];
  for (long long t = 1; t < M; t++) v = fmax(v, U[(size_t)t * N + i]);
  out[i] = v;
}

// out[t] = sum_i w[i] * U[t][i]: one 256-thread workgroup per saved state, fixed summation order
__global__ __launch_bounds__(256) void rowdot_kernel(int N, const double* __restrict__ U, const double* __restrict__ w,
                                                     double* __restrict__ out) {
  __shared__ double sh[4];
  const double* row = U + (size_t)blockIdx.x * N;
  double acc = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) acc += w[i] * row[i];
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// rates kernel with a skip flag (Newton iterations)
__global__ __launch_bounds__(256) void rates_skip_kernel(int R, const double* __restrict__ k, const double* __restrict__ u,
                                                         const int32_t* __restrict__ x0, const int32_t* __restrict__ x1,
                                                         double* __restrict__ rate, const int* skip) {
  const int sk = skip ? *skip : 0;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int32_t a = x0[r], b = x1[r];
  const double kr = k[r];
  if (sk) return;            // tested when the operand indices are back: a skipped launch never issues the gathers
  const double ub = b >= 0 ? u[b] : 1.0;
  rate[r] = kr * u[a] * ub;
}

// the same with the rate constants formed from a temperature and stored for the readers behind this launch (kernels.hpp: ArrheniusAt)
__global__ __launch_bounds__(256) void rates_skip_T_kernel(int R, ArrheniusAt at, double* __restrict__ k, const double* __restrict__ u,
                                                           const int32_t* __restrict__ x0, const int32_t* __restrict__ x1,
                                                           double* __restrict__ rate, const int* skip) {
  const int sk = skip ? *skip : 0;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int32_t a = x0[r], b = x1[r];
  // k is stored whether or not the iteration is skipped: the launches behind this one read it
  const double kr = arrhenius_one(at.Ea[r], at.A[r], 8.314462618 * at.T, at.has_kmax, at.k_max, at.t_mult);
  k[r] = kr;
  if (sk) return;
  const double ub = b >= 0 ? u[b] : 1.0;
  rate[r] = kr * u[a] * ub;
}

#define GRID1(n) dim3((unsigned)ceil_div((n), 256)), dim3(256)

void launch_bdf_predict(int N, int order, const double* D, const BdfCoef& cf, double atol, double rtol, double* y, double* psi,
                        double* d, double* scale, BdfCtrl* ctrl, hipStream_t s) {
  hipLaunchKernelGGL(bdf_predict_kernel, GRID1(N), 0, s, N, order, D, cf, atol, rtol, y, psi, d, scale, ctrl);
}
int bdf_reduce_blocks(int N) { return (int)ceil_div(N, RED_ELEMS); }
int bdf_reduce_slot() { return RED_SLOT; }
void launch_bdf_newton(int N, int iter, int maxit, double tol, const int32_t* xloc, const double* W, const double* scale,
                       double* y, double* d, double upd, double rate_max, double crate0, double tol_first, double dy_first_max,
                       int order, const double* D, double atol, double rtol, const BdfCoef& cf, BdfCtrl* ctrl, double* part,
                       BdfCtrl* host_ctrl, unsigned long long* host_seq, unsigned long long seq, bool publish_always, hipStream_t s,
                       bool crate_from_ctrl, bool ban_negatives) {
  hipLaunchKernelGGL(bdf_newton_kernel, dim3(bdf_reduce_blocks(N)), dim3(256), 0, s, N, iter, maxit, tol, xloc, W, scale, y, d, upd, rate_max,
                     crate0, tol_first, dy_first_max, order, D, atol, rtol, cf, ctrl, part, host_ctrl, host_seq, seq, publish_always ? 1 : 0,
                     crate_from_ctrl ? 1 : 0, ban_negatives ? 1 : 0);
}
void launch_bdf_accept_predict(int N, int ao, int order, double* D, const BdfCoef& cf, double atol, double rtol, double* y, double* psi,
                               double* d, double* scale, BdfCtrl* ctrl, double* copy_out, hipStream_t s, const int* go) {
  hipLaunchKernelGGL(bdf_accept_predict_kernel, GRID1(N), 0, s, N, ao, order, D, cf, atol, rtol, y, psi, d, scale, ctrl, copy_out, go);
}
void launch_bdf_accept(int N, int order, double* D, const double* d, double* copy_out, hipStream_t s) {
  hipLaunchKernelGGL(bdf_accept_kernel, GRID1(N), 0, s, N, order, D, d, copy_out);
}
void launch_clip_negative(int N, const double* a, double* out, hipStream_t s) {
  hipLaunchKernelGGL(clip_negative_kernel, GRID1(N), 0, s, N, a, out);
}
void launch_bdf_change_D(int N, int order, const BdfMat& ru, double* D, hipStream_t s) {
  hipLaunchKernelGGL(bdf_change_D_kernel, GRID1(N), 0, s, N, order, ru, D);
}
void launch_bdf_init_D(int N, int nrows, const double* y0, const double* f0, double h, double* D, hipStream_t s) {
  hipLaunchKernelGGL(bdf_init_D_kernel, GRID1(N), 0, s, N, nrows, y0, f0, h, D);
}
void launch_bdf_interp(int N, int order, const double* D, const BdfVec& p, double* out, hipStream_t s) {
  hipLaunchKernelGGL(bdf_interp_kernel, GRID1(N), 0, s, N, order, D, p, out);
}
void launch_rk_combine(int N, int n, const RkVec& w, const double* y, const double* K, double* out, hipStream_t s) {
  hipLaunchKernelGGL(rk_combine_kernel, GRID1(N), 0, s, N, n, w, y, K, out);
}
void launch_rk_error(int N, const RkVec& e, const double* y, const double* y_new, const double* K, double atol, double rtol,
                     BdfCtrl* ctrl, double* part, BdfCtrl* host_ctrl, unsigned long long* host_seq, unsigned long long seq,
                     hipStream_t s) {
  hipLaunchKernelGGL(rk_error_kernel, dim3(bdf_reduce_blocks(N)), dim3(256), 0, s, N, e, y, y_new, K, atol, rtol, ctrl, part, host_ctrl,
                     host_seq, seq);
}
void launch_axpy_out(int N, const double* a, double sc, const double* b, double* out, hipStream_t s) {
  hipLaunchKernelGGL(axpy_out_kernel, GRID1(N), 0, s, N, a, sc, b, out);
}
void launch_bdf_norms(int N, const double* y0, const double* f0, const double* f1, double atol, double rtol, BdfCtrl* ctrl, hipStream_t s) {
  hipLaunchKernelGGL(bdf_norms_kernel, dim3(1), dim3(1024), 0, s, N, y0, f0, f1, atol, rtol, ctrl);
}
void launch_colmax(int N, int64_t M, const double* U, double* out, hipStream_t s) {
  hipLaunchKernelGGL(colmax_kernel, GRID1(N), 0, s, N, (long long)M, U, out);
}
void launch_rowdot(int N, int64_t M, const double* U, const double* w, double* out, hipStream_t s) {
  if (M == 0) return;
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)M), dim3(256), 0, s, N, U, w, out);
}
void launch_rates_skip(int64_t R, const double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate,
                       const int* skip, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(rates_skip_kernel, GRID1(R), 0, s, (int)R, k, u, x0, x1, rate, skip);
}
void launch_rates_skip_T(int64_t R, const ArrheniusAt& at, double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate,
                         const int* skip, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(rates_skip_T_kernel, GRID1(R), 0, s, (int)R, at, k, u, x0, x1, rate, skip);
}

#include "ensemble_kernels.inc"

}  // namespace kin
