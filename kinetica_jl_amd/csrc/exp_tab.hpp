// Table-driven exp and the division-free Arrhenius evaluation shared by the rate-table kernels (kernels.hip,
// tiled_kernels.hip) and the sweep that forms its rate constants itself (tiled_kernels.hip, TMODE).
#pragma once
#include <hip/hip_runtime.h>

namespace kin {

// Arrhenius: k = A exp(-Ea/(R T)) N_A t_mult, optionally capped 1/(1/k_max + 1/k)
// (PrecalculatedArrheniusCalculator functor, src/solving/calculator.jl:223-232; constants.jl:4-5), evaluated literally
__device__ __forceinline__ double arrhenius_one(double Ea, double A, double RT, int has_kmax, double k_max, double t_mult) {
  const double kr = A * exp(-Ea / RT) * 6.02214076e23 * t_mult;
  // the cap in the reference's own form 1/(1/k_max + 1/k_r) (calculator.jl:225): exact limits at both ends
  // (k_r = inf -> k_max, k_r = 0 -> 0), where the cheaper k_r / (1 + k_r/k_max) gives NaN for an overflowing k_r
  return has_kmax ? 1.0 / (1.0 / k_max + 1.0 / kr) : kr;
}


// Table variant without IEEE divisions (the table kernel is FP64-VALU bound: with two full divisions and the library exp
// per element it ran at 2.0 ms for 14001 x 50000, against a 1.0 ms pure-store floor). Ea/RT is a multiply by the row's
// reciprocal plus one FMA residual correction; the cap 1/(1/k_max + 1/k_r) is evaluated in exactly that form,
// 1/k_r = exp(+Ea/RT) / (A N_A t_mult), with v_rcp_f64 and two Newton steps - an overflowing exp gives k = 0, the limit
// of the reference formula. Deviation from the two-division form: <= (2 |Ea/RT| + 8) * 2^-53 relative (one ulp in the
// argument of exp is amplified by |Ea/RT|), the bound the parity test applies element by element.
// Table-driven exp for the rate table (round 2): exp(x) = 2^m * T[j] * e^r with n = rint(x * 512/ln 2) = 512 m + j,
// r = x - n ln2/512 (two-part constant, n * hi exact: hi has 31 significant bits, |n| < 2^21 after the clamp),
// |r| <= ln2/1024 = 6.8e-4, e^r - 1 = r (1 + r (1/2 + r (1/6 + r/24))) (remainder r^5/120 < 1.3e-18 relative),
// T[j] = 2^(j/512) correctly rounded (exp2_tab.inc), kept in LDS. 11 FP64 operations instead of the 19 of exp_lean
// (degree-13 polynomial): the table kernel is FP64-VALU bound under sustained load. <= 1 ulp.
static __device__ const double kExp2Tab[512] = {
#include "exp2_tab.inc"
};

// TAB = 512: the table above. TAB = 128: every fourth entry of it (1 KB of LDS instead of 4: the tiled sweep keeps its
// whole state next to it), |r| <= ln2/256 = 2.7e-3 and one more term of the series (remainder r^6/720 < 6e-19).
template <int TAB>
__device__ __forceinline__ double exp_tab_t(double x, const double* __restrict__ tab_s) {
  static_assert(TAB == 512 || TAB == 128, "table sizes");
  constexpr double scale = TAB == 512 ? 0x1.71547652b82fep+9 : 0x1.71547652b82fep+7;      // TAB / ln 2
  constexpr double hi = TAB == 512 ? -0x1.62e42fec00000p-10 : -0x1.62e42fec00000p-8;      // ln 2 / TAB, 31 significant bits
  constexpr double lo = TAB == 512 ? -0x1.d1cf79abc9e3bp-41 : -0x1.d1cf79abc9e3bp-39;
  const double n = rint(x * scale);
  double r = fma(n, hi, x);
  r = fma(n, lo, r);
  const int ni = (int)n;
  const double T = tab_s[ni & (TAB - 1)];
  double p;
  if (TAB == 512) p = fma(r, 1.0 / 24.0, 1.0 / 6.0);
  else p = fma(fma(r, 1.0 / 120.0, 1.0 / 24.0), r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  return ldexp(fma(T, p * r, T), TAB == 512 ? ni >> 9 : ni >> 7);
}
__device__ __forceinline__ double exp_tab(double x, const double* __restrict__ tab_s) { return exp_tab_t<512>(x, tab_s); }

// c = A N_A t_mult, inv_c = 1 / c
template <int TAB>
__device__ __forceinline__ double arrhenius_fast_t(double Ea, double c, double inv_c, double RT, double inv_RT, int has_kmax,
                                                   double inv_kmax, const double* __restrict__ tab_s) {
  double q = Ea * inv_RT;
  q = fma(fma(-q, RT, Ea), inv_RT, q);
  q = fmin(q, 800.0);                                     // e^800 overflows anyway; keeps n inside the table arithmetic
  if (!has_kmax) return c * exp_tab_t<TAB>(-q, tab_s);
  const double x = fma(inv_c, exp_tab_t<TAB>(q, tab_s), inv_kmax);     // 1/k_max + 1/k_r
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return x < 1e300 ? y : 0.0;
}
__device__ __forceinline__ double arrhenius_fast(double Ea, double c, double inv_c, double RT, double inv_RT, int has_kmax,
                                                 double inv_kmax, const double* __restrict__ tab_s) {
  return arrhenius_fast_t<512>(Ea, c, inv_c, RT, inv_RT, has_kmax, inv_kmax, tab_s);
}

}  // namespace kin
