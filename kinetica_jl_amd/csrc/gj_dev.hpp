// Single-wavefront inversion of a 16x16 block by 16 dependent Gauss-Jordan steps (no pivoting, guarded), shared by the
// multi-workgroup dense inverse (solver_kernels.hip) and the resident integrator's in-workgroup one (resident.hip).
// Include from .hip files only.
#pragma once
#include <hip/hip_runtime.h>

namespace kin {

// A pivot counts as vanished when a multiplier exceeds PIVOT_GROWTH_MAX in magnitude (or is not finite), or when the pivot
// itself is below PIVOT_MIN in magnitude: the matrix is I - c J, whose natural scale is 1.
constexpr double PIVOT_GROWTH_MAX = 1e8;
constexpr double PIVOT_MIN = 1e-8;

__device__ __forceinline__ double fast_recip(double x) {
  double r = __builtin_amdgcn_rcp(x);     // v_rcp_f64, then two Newton steps (quadratic: full precision)
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}

// Inverse of the 32x32 pivot block (the critical path of every gj_update_kernel launch). Two levels: the block is split
// into 16x16 quarters, inv([[A11, A12], [A21, A22]]) = [[B11, B12], [B21, B22]] with S = A22 - A21 A11^-1 A12,
//   B22 = S^-1,  B12 = -(A11^-1 A12) B22,  B21 = -B22 (A21 A11^-1),  B11 = A11^-1 - B12 (A21 A11^-1).
// The two 16x16 inversions are 16 dependent Gauss-Jordan steps each inside ONE wavefront (lane = row r, four consecutive
// columns in registers; pivot row / column / pivot travel by lane exchanges, no workgroup barrier per step); the six
// 16x16x16 products in between are matrix-core tiles of the same wavefront. 32 dependent steps either way, but a step
// costs a few lane exchanges instead of an LDS round trip plus a workgroup barrier (the one-level version, thread =
// (row, 4 columns) of the whole block, took ~7 us of each 13 us block step).
// The block is in A on entry (XS and, once read, A serve as scratch); the result goes to `pinv`.
// Lane exchanges of gj_inv16_wave that do not need the LDS crossbar: a wave-uniform source lane (v_readlane), and the
// broadcast of lane Q of every quad to its quad (DPP quad_perm).
__device__ __forceinline__ double gj_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int Q>
__device__ __forceinline__ double gj_quad_bcast(double v) {
  constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Step K of the 16 dependent Gauss-Jordan steps. Every lane carries a private copy `rk` of "its" four columns of the
// NEXT pivot row, kept up to date with the same operations the row's owner lanes apply, so that the only things on the
// step-to-step critical path are the pivot's reciprocal and two multiply-adds; the lane shuffle that fetches row K + 1
// (ds_bpermute, ~100 cycles) is issued a whole step before its result is needed, the pivot and the multipliers travel
// by v_readlane / DPP. (First version: pivot, multiplier and pivot row all by ds_bpermute after the update, then the
// reciprocal: ~250 cycles per step, 5.6 us per 32x32 block - tools/gj_probe.hip.) The arithmetic, and with it the
// result, is bit for bit that of the first version.
template <int K>
__device__ __forceinline__ void gj_step16(double (&a)[4], double (&rk)[4], bool& vanished, int lane) {
  const int r = lane >> 2, q = lane & 3;
  constexpr int QK = K >> 2, JK = K & 3;       // column K: lane group QK, register JK
  // the pivot: every lane of group QK holds it in its copy of the pivot row
  const double piv = gj_readlane(rk[JK], QK);
  // next pivot row as it is now (before this step's update), and its multiplier
  double nr[4] = {0.0, 0.0, 0.0, 0.0};
  double m = 0.0;
  if (K < 15) {
#pragma unroll
    for (int j = 0; j < 4; j++) nr[j] = __shfl(a[j], 4 * (K + 1) + q, 64);
    m = gj_readlane(a[JK], 4 * (K + 1) + QK);
  }
  const double aik = gj_quad_bcast<QK>(a[JK]);   // my row's element of the pivot column
  const double inv = fast_recip(piv);
  // vanished pivot: the multiplier of this row exceeds the growth bound, or the pivot is tiny / 0 / not finite
  vanished |= (r != K) ? (aik != 0.0 && !(fabs(aik * inv) <= PIVOT_GROWTH_MAX)) : !(fabs(piv) >= PIVOT_MIN);
  // (a single wavefront issues this chain: the instruction count per step is what it costs - only register JK can
  // hold the pivot column, the other three need no selects for it)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (j == JK) {
      const bool kc = q == QK;
      const double rkj = (kc ? 1.0 : rk[j]) * inv;
      a[j] = (r == K) ? rkj : (kc ? 0.0 : a[j]) - aik * rkj;
      nr[j] = (kc ? 0.0 : nr[j]) - m * rkj;
    } else {
      const double rkj = rk[j] * inv;
      a[j] = (r == K) ? rkj : a[j] - aik * rkj;
      nr[j] = nr[j] - m * rkj;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) rk[j] = nr[j];
}

__device__ __forceinline__ bool gj_inv16_wave(double (&a)[4], int lane) {
  bool vanished = false;
  double rk[4];
#pragma unroll
  for (int j = 0; j < 4; j++) rk[j] = __shfl(a[j], lane & 3, 64);   // row 0
  gj_step16<0>(a, rk, vanished, lane);   gj_step16<1>(a, rk, vanished, lane);   gj_step16<2>(a, rk, vanished, lane);
  gj_step16<3>(a, rk, vanished, lane);   gj_step16<4>(a, rk, vanished, lane);   gj_step16<5>(a, rk, vanished, lane);
  gj_step16<6>(a, rk, vanished, lane);   gj_step16<7>(a, rk, vanished, lane);   gj_step16<8>(a, rk, vanished, lane);
  gj_step16<9>(a, rk, vanished, lane);   gj_step16<10>(a, rk, vanished, lane);  gj_step16<11>(a, rk, vanished, lane);
  gj_step16<12>(a, rk, vanished, lane);  gj_step16<13>(a, rk, vanished, lane);  gj_step16<14>(a, rk, vanished, lane);
  gj_step16<15>(a, rk, vanished, lane);
  return vanished;
}

}  // namespace kin
