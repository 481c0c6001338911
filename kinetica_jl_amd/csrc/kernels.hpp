// Device-side tables and kernel launchers of libkinetica_hip (gfx950 only).
#pragma once
#include "common.hpp"
#include "network.hpp"

namespace kin {

// raw-pointer view of a SegPlan, passed to kernels by value
struct SegPlanView {
  const int32_t* grp_off; const int32_t* grp_dst; const int32_t* grp_aux;
  const int32_t* ell_a; const int32_t* ell_b; const float* ell_c;
  const int32_t* seg_beg; const int32_t* seg_end; const int32_t* seg_dst; const int32_t* seg_aux;
  const int32_t* blk_beg; const int32_t* blk_end; const int32_t* blk_dst; const int32_t* blk_aux;
  const int32_t* long_a; const int32_t* long_b; const float* long_c;
  int32_t G, S, B;
  int32_t val_base, ell_total;   // value-ordered product plans (SegPlanHost::val_base), else val_base < 0
};

struct SegPlanDev {
  DevBuf<int32_t> grp_off, grp_dst, grp_aux, ell_a, ell_b, seg_beg, seg_end, seg_dst, seg_aux, blk_beg, blk_end, blk_dst, blk_aux, long_a, long_b;
  DevBuf<float> ell_c, long_c;
  int32_t G = 0, S = 0, B = 0, val_base = -1, ell_total = 0;
  void upload(const SegPlanHost& h, hipStream_t s);
  SegPlanView view() const;
};

// what an entry contributes and how the row sum is combined with the output
enum SegOp : int {
  SEG_COEF_SET = 0,   // out[dst]  = sum c * src[a]
  SEG_PROD_SUB = 1,   // out[dst] -= sum src[a] * src[b]                 (Schur update of the sparse LU)
  SEG_COEF_BDF = 2,   // out[dst]  = cscal * (sum c * src[a]) - psi[aux] - d[aux]   (Newton residual of a BDF step;
                      //            aux = species index, dst = position in the permuted solve vector)
  SEG_PROD_SUB_DIV = 3,  // out[dst] = (out[dst] - sum src[a] * src[b]) / src[aux]        (backward substitution)
  SEG_PROD_AUXSUB = 4,   // out[dst] = src[aux] - sum src[a] * src[b]     (explicit triangular inverses: y1 = b1 - Z' b1, t = y1 - U12 x2)
  SEG_PROD_SET = 5,      // out[dst] = sum src[a] * src[b]                 (x1 = V t)
  SEG_PROD_NEG = 6,      // out[dst] = - sum src[a] * src[b]               (NVU = -(V U12) of the fused solve)
};
struct SegExtra {  // extra operands of SEG_COEF_BDF
  const double* psi = nullptr; const double* d = nullptr; double cscal = 0.0;
  const int* skip = nullptr;   // optional device flag: the kernel returns immediately when *skip != 0
};

void launch_segsum(const SegPlanView& p, SegOp op, const double* src, double* out, const SegExtra& ex, hipStream_t s);

// per-reaction rate and operand derivatives (single state)
void launch_rates(int64_t R, const double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate, hipStream_t s);
void launch_drates(int64_t R, const double* k, const double* u, const int32_t* x0, const int32_t* x1, double* dr, hipStream_t s);

// a temperature whose rate constants have not been formed yet (continuous-rate solves): passed by value to the kernels
// that evaluate the Arrhenius law themselves and store k for the readers behind them
struct ArrheniusAt { const double* Ea; const double* A; int has_kmax; double k_max, t_mult, T; };
void launch_rates_T(int64_t R, const ArrheniusAt& at, double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate, hipStream_t s);
void launch_drates_T(int64_t R, const ArrheniusAt& at, double* k, const double* u, const int32_t* x0, const int32_t* x1, double* dr, hipStream_t s);

// Arrhenius (k_max < 0 or NaN handled by has_kmax flag)
void launch_arrhenius(int64_t n, const double* Ea, const double* A, int has_kmax, double k_max, double t_mult, double T, double* k, hipStream_t s);
void launch_rate_table(int64_t n, int64_t n_stops, const double* Ea, const double* A, int has_kmax, double k_max,
                       double t_mult, const double* T, double* table, hipStream_t s);

// batched sweep over B states, state-major layouts (see kin_rhs_batched_dev); rec = packed 16-byte records
// `adjacent`: pair p = reactions (2p, 2p+1); `block`: pair p = reactions (p, P+p) (forwards first, reverses behind)
// n_cu: compute units of the device the handle lives on (per handle, not cached per process)
void launch_sweep(int n_cu, int64_t N, int64_t R, int64_t P, int64_t B, bool adjacent, bool block, const void* rec, const void* pair_k,
                  const void* rec64, const int32_t* copy_species, int n_copy, const void* gen_rec8, const int32_t* gen_expl,
                  int n_gen_expl, const double* u, const double* k_b, const double* k_1, double* du, hipStream_t s);

// large-N sweep (state too large for LDS): hubs in LDS, tail via a per-workgroup scratch row and tail-entry lists;
// `scratch` holds min(B, n_cu) rows of (N - H) + P doubles
void launch_sweep_big(int n_cu, int64_t N, int64_t R, int64_t P, int64_t B, bool adjacent, int32_t H, int32_t n_tail_tiles, const void* rec8,
                      const void* rec, const int32_t* expl, int32_t n_expl, const void* pair_k, const int32_t* spec_of_label,
                      const int32_t* tail_ptr, const void* tail_ent, double* scratch, const double* u, const double* k_b,
                      const double* k_1, double* du, bool tail_by_species, hipStream_t s);

}  // namespace kin
