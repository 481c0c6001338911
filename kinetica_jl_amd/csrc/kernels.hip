// Hand-written gfx950 (CDNA4, wave64) kernels for the mass-action RHS, its Jacobian, the
// Arrhenius rate table and the batched RHS sweep. All FP64; every kernel is memory/latency
// bound (arithmetic intensity ~0.25 flop/B), so the rules that matter are coalescing,
// wave-uniform (scalar) index loads, fixed summation order and enough waves in flight.
#include "kernels.hpp"

namespace kin {

// ------------------------------------------------------------------------------------------
// plan upload
// ------------------------------------------------------------------------------------------
void SegPlanDev::upload(const SegPlanHost& h, hipStream_t s) {
  grp_off.upload(h.grp_off, s); grp_dst.upload(h.grp_dst, s); grp_aux.upload(h.grp_aux, s);
  ell_a.upload(h.ell_a, s); ell_b.upload(h.ell_b, s); ell_c.upload(h.ell_c, s);
  seg_beg.upload(h.seg_beg, s); seg_end.upload(h.seg_end, s); seg_dst.upload(h.seg_dst, s); seg_aux.upload(h.seg_aux, s);
  long_a.upload(h.long_a, s); long_b.upload(h.long_b, s); long_c.upload(h.long_c, s);
  fix_dst.upload(h.fix_dst, s); fix_aux.upload(h.fix_aux, s); fix_ptr.upload(h.fix_ptr, s);
  partials.alloc((size_t)h.n_partials + 1);
  G = h.n_groups(); S = h.n_segs(); F = h.n_fix();
  KIN_HIP(hipStreamSynchronize(s));  // host vectors may die after this call
}

SegPlanView SegPlanDev::view() const {
  return SegPlanView{grp_off.p, grp_dst.p, grp_aux.p, ell_a.p, ell_b.p, ell_c.p, seg_beg.p, seg_end.p, seg_dst.p, seg_aux.p,
                     long_a.p, long_b.p, long_c.p, fix_dst.p, fix_aux.p, fix_ptr.p, partials.p, G, S, F};
}

// ------------------------------------------------------------------------------------------
// deterministic segmented gather-sum
// ------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ void seg_store(double* out, const double* src, int32_t dst, int32_t aux, double acc,
                                          const SegExtra& ex) {
  if (OP == SEG_COEF_SET) out[dst] = acc;
  else if (OP == SEG_PROD_SUB) out[dst] -= acc;
  else if (OP == SEG_PROD_SUB_DIV) out[dst] = (out[dst] - acc) / src[aux];
  else out[dst] = ex.cscal * acc - ex.psi[aux] - ex.d[aux];
}
template <int OP> struct seg_is_prod { static constexpr bool v = (OP == SEG_PROD_SUB || OP == SEG_PROD_SUB_DIV); };

__device__ __forceinline__ double wave_sum(double v) {
  // fixed butterfly order -> bitwise reproducible
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <int OP>
__global__ __launch_bounds__(256) void segsum_kernel(SegPlanView p, const double* src, double* out, SegExtra ex) {
  if (ex.skip && *ex.skip) return;
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task < p.G) {
    const int32_t dst = p.grp_dst[task * 64 + lane];
    const int32_t c0 = p.grp_off[task], c1 = p.grp_off[task + 1];
    double acc = 0.0;
    for (int32_t col = c0; col < c1; col++) {
      const size_t idx = (size_t)col * 64 + lane;
      const float c = p.ell_c[idx];
      if (c != 0.0f) {
        if (seg_is_prod<OP>::v) acc += src[p.ell_a[idx]] * src[p.ell_b[idx]];
        else acc += (double)c * src[p.ell_a[idx]];
      }
    }
    if (dst >= 0) seg_store<OP>(out, src, dst, p.grp_aux[task * 64 + lane], acc, ex);
  } else if (task < p.G + p.S) {
    const int sidx = task - p.G;
    const int32_t e0 = p.seg_beg[sidx], e1 = p.seg_end[sidx];
    double acc = 0.0;
    for (int32_t e = e0 + lane; e < e1; e += 64) {
      if (seg_is_prod<OP>::v) acc += src[p.long_a[e]] * src[p.long_b[e]];
      else acc += (double)p.long_c[e] * src[p.long_a[e]];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const int32_t dst = p.seg_dst[sidx];
      if (dst >= 0) seg_store<OP>(out, src, dst, p.seg_aux[sidx], acc, ex);
      else p.partials[-dst - 1] = acc;
    }
  }
}

template <int OP>
__global__ void segsum_fix_kernel(SegPlanView p, const double* src, double* out, SegExtra ex) {
  if (ex.skip && *ex.skip) return;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= p.F) return;
  double acc = 0.0;
  for (int32_t q = p.fix_ptr[f]; q < p.fix_ptr[f + 1]; q++) acc += p.partials[q];
  seg_store<OP>(out, src, p.fix_dst[f], p.fix_aux[f], acc, ex);
}

void launch_segsum(const SegPlanView& p, SegOp op, const double* src, double* out, const SegExtra& ex, hipStream_t s) {
  const int tasks = p.G + p.S;
  if (tasks > 0) {
    dim3 grid((unsigned)ceil_div(tasks, 4)), block(256);
    switch (op) {
      case SEG_COEF_SET: hipLaunchKernelGGL(segsum_kernel<SEG_COEF_SET>, grid, block, 0, s, p, src, out, ex); break;
      case SEG_PROD_SUB: hipLaunchKernelGGL(segsum_kernel<SEG_PROD_SUB>, grid, block, 0, s, p, src, out, ex); break;
      case SEG_COEF_BDF: hipLaunchKernelGGL(segsum_kernel<SEG_COEF_BDF>, grid, block, 0, s, p, src, out, ex); break;
      case SEG_PROD_SUB_DIV: hipLaunchKernelGGL(segsum_kernel<SEG_PROD_SUB_DIV>, grid, block, 0, s, p, src, out, ex); break;
    }
  }
  if (p.F > 0) {
    dim3 grid((unsigned)ceil_div(p.F, 64)), block(64);
    switch (op) {
      case SEG_COEF_SET: hipLaunchKernelGGL(segsum_fix_kernel<SEG_COEF_SET>, grid, block, 0, s, p, src, out, ex); break;
      case SEG_PROD_SUB: hipLaunchKernelGGL(segsum_fix_kernel<SEG_PROD_SUB>, grid, block, 0, s, p, src, out, ex); break;
      case SEG_COEF_BDF: hipLaunchKernelGGL(segsum_fix_kernel<SEG_COEF_BDF>, grid, block, 0, s, p, src, out, ex); break;
      case SEG_PROD_SUB_DIV: hipLaunchKernelGGL(segsum_fix_kernel<SEG_PROD_SUB_DIV>, grid, block, 0, s, p, src, out, ex); break;
    }
  }
  KIN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// per-reaction rates and operand derivatives (make_rs mass action, solve_utils.jl:318-334)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rates_kernel(int R, const double* __restrict__ k, const double* __restrict__ u,
                                                    const int32_t* __restrict__ x0, const int32_t* __restrict__ x1,
                                                    double* __restrict__ rate) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int32_t a = x0[r], b = x1[r];
  const double ub = b >= 0 ? u[b] : 1.0;
  rate[r] = k[r] * u[a] * ub;
}

// dr[2r]   = d rate_r / d u[x0]   (2A: 2 k u, single column)
// dr[2r+1] = d rate_r / d u[x1]   (only for A + B)
__global__ __launch_bounds__(256) void drates_kernel(int R, const double* __restrict__ k, const double* __restrict__ u,
                                                     const int32_t* __restrict__ x0, const int32_t* __restrict__ x1,
                                                     double* __restrict__ dr) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int32_t a = x0[r], b = x1[r];
  const double kk = k[r];
  double d0, d1 = 0.0;
  if (b < 0) d0 = kk;
  else if (b == a) d0 = 2.0 * kk * u[a];
  else { d0 = kk * u[b]; d1 = kk * u[a]; }
  reinterpret_cast<double2*>(dr)[r] = make_double2(d0, d1);
}

void launch_rates(int64_t R, const double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(rates_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s, (int)R, k, u, x0, x1, rate);
  KIN_HIP(hipGetLastError());
}
void launch_drates(int64_t R, const double* k, const double* u, const int32_t* x0, const int32_t* x1, double* dr, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(drates_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s, (int)R, k, u, x0, x1, dr);
  KIN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// Arrhenius: k = A exp(-Ea/(R T)) N_A t_mult, optionally capped 1/(1/k_max + 1/k)
// (PrecalculatedArrheniusCalculator functor, src/solving/calculator.jl:223-232; constants.jl:4-5)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double arrhenius_one(double Ea, double A, double RT, int has_kmax, double inv_kmax, double t_mult) {
  const double kr = A * exp(-Ea / RT) * 6.02214076e23 * t_mult;
  return has_kmax ? 1.0 / (inv_kmax + (1.0 / kr)) : kr;
}

__global__ __launch_bounds__(256) void arrhenius_kernel(int n, const double* __restrict__ Ea, const double* __restrict__ A,
                                                        int has_kmax, double k_max, double t_mult, double T,
                                                        double* __restrict__ k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  k[i] = arrhenius_one(Ea[i], A[i], 8.314462618 * T, has_kmax, 1.0 / k_max, t_mult);
}

// table[s][r]; one thread produces two consecutive reactions (16-byte stores), grid.y walks
// time stops so that each workgroup keeps its (Ea, A) pairs in registers across ROWS_PER_BLOCK rows.
constexpr int TABLE_ROWS_PER_BLOCK = 8;
__global__ __launch_bounds__(256) void rate_table_kernel(int n, int n_stops, const double* __restrict__ Ea,
                                                         const double* __restrict__ A, int has_kmax, double k_max,
                                                         double t_mult, const double* __restrict__ T,
                                                         double* __restrict__ table) {
  const int r = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (r >= n) return;
  const int s0 = blockIdx.y * TABLE_ROWS_PER_BLOCK;
  const int s1 = min(n_stops, s0 + TABLE_ROWS_PER_BLOCK);
  const bool pair = (r + 1 < n);
  const double e0 = Ea[r], a0 = A[r];
  const double e1 = pair ? Ea[r + 1] : 0.0, a1 = pair ? A[r + 1] : 1.0;
  const double inv_kmax = 1.0 / k_max;
  for (int s = s0; s < s1; s++) {
    const double RT = 8.314462618 * T[s];
    const double v0 = arrhenius_one(e0, a0, RT, has_kmax, inv_kmax, t_mult);
    double* row = table + (size_t)s * n;
    if (pair && ((n & 1) == 0)) {
      const double v1 = arrhenius_one(e1, a1, RT, has_kmax, inv_kmax, t_mult);
      *reinterpret_cast<double2*>(row + r) = make_double2(v0, v1);
    } else {
      row[r] = v0;
      if (pair) row[r + 1] = arrhenius_one(e1, a1, RT, has_kmax, inv_kmax, t_mult);
    }
  }
}

void launch_arrhenius(int64_t n, const double* Ea, const double* A, int has_kmax, double k_max, double t_mult, double T, double* k, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(arrhenius_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, (int)n, Ea, A, has_kmax, k_max, t_mult, T, k);
  KIN_HIP(hipGetLastError());
}
void launch_rate_table(int64_t n, int64_t n_stops, const double* Ea, const double* A, int has_kmax, double k_max,
                       double t_mult, const double* T, double* table, hipStream_t s) {
  if (n == 0 || n_stops == 0) return;
  dim3 grid((unsigned)ceil_div(n, 512), (unsigned)ceil_div(n_stops, TABLE_ROWS_PER_BLOCK));
  hipLaunchKernelGGL(rate_table_kernel, grid, dim3(256), 0, s, (int)n, (int)n_stops, Ea, A, has_kmax, k_max, t_mult, T, table);
  KIN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// batched sweep, two passes over state-contiguous arrays (lanes = states, 2 states per lane)
//   pass 1: rate[r][b] = k[r][b] * u[x0][b] * u[x1][b]       (reaction-major, streaming)
//   pass 2: du[i][b]   = sum_e coef_e * rate[rxn_e][b]       (species-major, deterministic gather)
// Reaction / species indices are wave-uniform, so they travel through the scalar cache.
// ------------------------------------------------------------------------------------------
constexpr int RB_RXN_PER_WAVE = 8;
__global__ __launch_bounds__(256) void rates_batched_kernel(int R, int ldb, const double* __restrict__ k_rb,
                                                            const double* __restrict__ k_r, const double* __restrict__ u,
                                                            const int32_t* __restrict__ x0, const int32_t* __restrict__ x1,
                                                            double* __restrict__ rate) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.y * 64 + lane) * 2;            // two states per lane (16-byte accesses)
  if (b >= ldb) return;
  const int r0 = (blockIdx.x * 4 + wave) * RB_RXN_PER_WAVE;
#pragma unroll
  for (int j = 0; j < RB_RXN_PER_WAVE; j++) {
    const int r = r0 + j;
    if (r >= R) break;
    const int32_t a = x0[r], c = x1[r];  // uniform -> s_load
    double2 kv = k_rb ? *reinterpret_cast<const double2*>(k_rb + (size_t)r * ldb + b) : make_double2(k_r[r], k_r[r]);
    const double2 ua = *reinterpret_cast<const double2*>(u + (size_t)a * ldb + b);
    double2 out = make_double2(kv.x * ua.x, kv.y * ua.y);
    if (c >= 0) {
      const double2 uc = *reinterpret_cast<const double2*>(u + (size_t)c * ldb + b);
      out.x *= uc.x; out.y *= uc.y;
    }
    *reinterpret_cast<double2*>(rate + (size_t)r * ldb + b) = out;
  }
}

// one wavefront per (species, 128-state tile); long rows are walked by all 4 waves of the
// workgroup and combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void gather_batched_kernel(int N, int ldb, const int32_t* __restrict__ sp_ptr,
                                                             const int32_t* __restrict__ sp_rxn,
                                                             const float* __restrict__ sp_coef,
                                                             const int32_t* __restrict__ row_order,
                                                             const double* __restrict__ rate, double* __restrict__ du) {
  __shared__ double2 red[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.y * 64 + lane) * 2;
  const bool active = b < ldb;
  // blockIdx.x enumerates rows in `row_order` (longest first so hub rows start early)
  const int i = row_order[blockIdx.x];
  const int32_t e0 = sp_ptr[i], e1 = sp_ptr[i + 1];
  double2 acc = make_double2(0.0, 0.0);
  if (active) {
    for (int32_t e = e0 + wave; e < e1; e += 4) {
      const int32_t r = sp_rxn[e];
      const double c = (double)sp_coef[e];
      const double2 v = *reinterpret_cast<const double2*>(rate + (size_t)r * ldb + b);
      acc.x += c * v.x; acc.y += c * v.y;
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && active) {
    double2 t = red[0][lane];
#pragma unroll
    for (int w = 1; w < 4; w++) { t.x += red[w][lane].x; t.y += red[w][lane].y; }
    *reinterpret_cast<double2*>(du + (size_t)i * ldb + b) = t;
  }
}

void launch_rates_batched(int64_t R, int64_t B, int64_t ldb, const double* k_rb, const double* k_r, const double* u,
                          const int32_t* x0, const int32_t* x1, double* rate, hipStream_t s) {
  (void)B;
  if (R == 0) return;
  dim3 grid((unsigned)ceil_div(R, 4 * RB_RXN_PER_WAVE), (unsigned)ceil_div(ldb, 128));
  hipLaunchKernelGGL(rates_batched_kernel, grid, dim3(256), 0, s, (int)R, (int)ldb, k_rb, k_r, u, x0, x1, rate);
  KIN_HIP(hipGetLastError());
}
void launch_gather_batched(int64_t N, int64_t B, int64_t ldb, const int32_t* sp_ptr, const int32_t* sp_rxn,
                           const float* sp_coef, const int32_t* row_order, const double* rate, double* du, hipStream_t s) {
  (void)B;
  dim3 grid((unsigned)N, (unsigned)ceil_div(ldb, 128));
  hipLaunchKernelGGL(gather_batched_kernel, grid, dim3(256), 0, s, (int)N, (int)ldb, sp_ptr, sp_rxn, sp_coef, row_order, rate, du);
  KIN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// tiled transpose (layout conversion of the host-buffer batched API)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const double* __restrict__ in, int ld_in,
                                                        double* __restrict__ out, int ld_out) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = in[(size_t)r * ld_in + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) out[(size_t)c * ld_out + r] = tile[tx][j];
  }
}

void launch_transpose(int64_t rows, int64_t cols, const double* in, int64_t ld_in, double* out, int64_t ld_out, hipStream_t s) {
  if (rows == 0 || cols == 0) return;
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, (int)rows, (int)cols, in, (int)ld_in, out, (int)ld_out);
  KIN_HIP(hipGetLastError());
}

}  // namespace kin
