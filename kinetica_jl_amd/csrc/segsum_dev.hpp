// Device-side pieces of the deterministic segmented gather-sum (kernels.hip: segsum_kernel; solver_kernels.hip: the
// solve's last gather stage fused with the corrector update). Include from .hip files only.
#pragma once
#include "kernels.hpp"

namespace kin {

// The store operands that do not depend on the sum (old value, pivot, psi, d) are loaded by seg_pre
// BEFORE the gather loop: these kernels are a chain of dependent global loads on a few thousand rows,
// so every load taken off the critical path is ~1 us per launch. The same goes for the skip flag
// (blind-enqueued Newton iterations): it is loaded together with the row descriptors and tested when those
// are back - a skipped launch (47 % of the second iterations at C3) ends after ONE round of loads instead of
// three, an active one does not wait for the flag any longer than it waits for its descriptors anyway.
// (The pointer and plan-view types are template parameters: the grid kernels pass plain pointers and SegPlanView, the resident
// integrator address-space-qualified ones - pointers read from a context block are generic to the compiler and would turn
// every access into a flat instruction.)
struct SegPre { double o, a, b; };
template <int OP, class OutP, class SrcP, class EX>
__device__ __forceinline__ SegPre seg_pre(OutP out, SrcP src, int32_t dst, int32_t aux, const EX& ex) {
  SegPre q{0.0, 0.0, 0.0};
  if (dst < 0) return q;
  if (OP == SEG_PROD_SUB) q.o = out[dst];
  else if (OP == SEG_PROD_SUB_DIV) { q.o = out[dst]; q.a = src[aux]; }
  else if (OP == SEG_PROD_AUXSUB) q.o = src[aux];
  else if (OP == SEG_COEF_BDF) { q.a = ex.psi[aux]; q.b = ex.d[aux]; }
  return q;
}
template <int OP, class OutP, class EX>
__device__ __forceinline__ void seg_store(OutP out, int32_t dst, double acc, const SegPre& q, const EX& ex) {
  if (OP == SEG_COEF_SET) out[dst] = acc;
  else if (OP == SEG_PROD_SUB) out[dst] = q.o - acc;
  else if (OP == SEG_PROD_SUB_DIV) out[dst] = (q.o - acc) / q.a;
  else if (OP == SEG_PROD_AUXSUB) out[dst] = q.o - acc;
  else if (OP == SEG_PROD_SET) out[dst] = acc;
  else if (OP == SEG_PROD_NEG) out[dst] = -acc;
  else out[dst] = ex.cscal * acc - q.a - q.b;
}
template <int OP> struct seg_is_prod { static constexpr bool v = (OP == SEG_PROD_SUB || OP == SEG_PROD_SUB_DIV || OP == SEG_PROD_AUXSUB || OP == SEG_PROD_SET || OP == SEG_PROD_NEG); };

__device__ __forceinline__ double wave_sum(double v) {
  // fixed butterfly order -> bitwise reproducible
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One round of NX entries per lane: all index loads, then all gathers, then the sum in slot order (two dependent loads on
// the critical path). `idx_of(x)` = payload index of this lane's x-th entry, or -1. ELL = entries of an ELL group (padding
// inside the group is marked in the payload), else the contiguous payload of medium / long rows.
// srcA: the array the first factor (or the coefficient plans' operand) is read from, srcB: the second factor's - the same
// array in the grid kernels; in the resident integrator the factor values are in global memory and the solve vectors in LDS
template <int OP, int NX, bool ELL, class View, class SrcA, class SrcB, class EX, class F>
__device__ __forceinline__ double seg_gather2(const View& p, SrcA srcA, SrcB srcB, const EX& ex, bool impl, F idx_of) {
  constexpr bool PROD = seg_is_prod<OP>::v;
  const auto A = ELL ? p.ell_a : p.long_a;
  const auto Bp = ELL ? p.ell_b : p.long_b;
  const auto C = ELL ? p.ell_c : p.long_c;
  const int32_t vbase = ELL ? p.val_base : p.val_base + p.ell_total;
  float c[NX]; int32_t ia[NX], ib[NX]; double va[NX], vb[NX];
#pragma unroll
  for (int x = 0; x < NX; x++) {
    const int32_t e = idx_of(x);
    const bool ok = e >= 0;
    ib[x] = (PROD && ok) ? Bp[e] : -1;
    if (PROD && impl) ia[x] = vbase + e; else ia[x] = ok ? A[e] : 0;
    c[x] = PROD ? (ib[x] >= 0 ? 1.0f : 0.0f) : (ok ? C[e] : 0.0f);   // product plans carry no coefficients: padding = b < 0
  }
#pragma unroll
  for (int x = 0; x < NX; x++) {
    const bool on = c[x] != 0.0f;
    va[x] = on ? srcA[ia[x]] : 0.0;
    vb[x] = (PROD && on) ? srcB[ib[x]] : 0.0;
  }
  double acc = 0.0;
#pragma unroll
  for (int x = 0; x < NX; x++) {
    if (PROD) acc += va[x] * vb[x];
    else acc += (double)c[x] * va[x];
  }
  return acc;
}

template <int OP, int NX, bool ELL, class View, class SrcP, class EX, class F>
__device__ __forceinline__ double seg_gather(const View& p, SrcP src, const EX& ex, bool impl, F idx_of) {
  return seg_gather2<OP, NX, ELL>(p, src, src, ex, impl, idx_of);
}

}  // namespace kin
