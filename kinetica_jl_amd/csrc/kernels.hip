// Hand-written gfx950 (CDNA4, wave64) kernels for the mass-action RHS, its Jacobian, the
// Arrhenius rate table and the batched RHS sweep. All FP64; every kernel is memory/latency
// bound (arithmetic intensity ~0.25 flop/B), so the rules that matter are coalescing,
// wave-uniform (scalar) index loads, fixed summation order and enough waves in flight.
#include "kernels.hpp"

#include "exp_tab.hpp"
#include "segsum_dev.hpp"

#include <algorithm>

namespace kin {

// ------------------------------------------------------------------------------------------
// plan upload
// ------------------------------------------------------------------------------------------
void SegPlanDev::upload(const SegPlanHost& h, hipStream_t s) {
  grp_off.upload(h.grp_off, s); grp_dst.upload(h.grp_dst, s); grp_aux.upload(h.grp_aux, s);
  ell_a.upload(h.ell_a, s); ell_b.upload(h.ell_b, s); ell_c.upload(h.ell_c, s);
  seg_beg.upload(h.seg_beg, s); seg_end.upload(h.seg_end, s); seg_dst.upload(h.seg_dst, s); seg_aux.upload(h.seg_aux, s);
  long_a.upload(h.long_a, s); long_b.upload(h.long_b, s); long_c.upload(h.long_c, s);
  blk_beg.upload(h.blk_beg, s); blk_end.upload(h.blk_end, s); blk_dst.upload(h.blk_dst, s); blk_aux.upload(h.blk_aux, s);
  G = h.n_groups(); S = h.n_segs(); B = h.n_blks(); val_base = h.val_base; ell_total = h.ell_total;
  KIN_HIP(hipStreamSynchronize(s));  // host vectors may die after this call
}

SegPlanView SegPlanDev::view() const {
  return SegPlanView{grp_off.p, grp_dst.p, grp_aux.p, ell_a.p, ell_b.p, ell_c.p, seg_beg.p, seg_end.p, seg_dst.p, seg_aux.p,
                     blk_beg.p, blk_end.p, blk_dst.p, blk_aux.p, long_a.p, long_b.p, long_c.p, G, S, B, val_base, ell_total};
}

// ------------------------------------------------------------------------------------------
// deterministic segmented gather-sum
// ------------------------------------------------------------------------------------------
// grid of 1024-thread workgroups: the first p.B take one LONG row each (whole workgroup, BLK_PASS entries per pass, so
// that all but the very longest rows are ONE round of index loads + gathers), the others sixteen wavefront tasks each
// (an ELL group of 64 short rows, or one medium row)
// (plans without long rows are launched with 256-thread workgroups: small workgroups start ~1.5 us sooner)
template <int OP, int SEG_WG>
__global__ __launch_bounds__(SEG_WG) void segsum_kernel(SegPlanView p, const double* src, double* out, SegExtra ex) {
  constexpr int SEG_WAVES = SEG_WG / 64, BLK_PER_THREAD = SegPlanHost::BLK_PASS / 1024;
  const int skip = ex.skip ? *ex.skip : 0;
  const int lane = threadIdx.x & 63;
  const bool impl = p.val_base >= 0;                    // value-ordered plan: first factor of slot q = src[val_base + q]
  if (SEG_WG == 1024 && (int)blockIdx.x < p.B) {
    __shared__ double sh[SEG_WAVES];
    const int r = blockIdx.x;
    const int32_t e0 = p.blk_beg[r], e1 = p.blk_end[r];
    const int32_t bdst = p.blk_dst[r];
    const int32_t baux = p.blk_aux[r];
    if (skip) return;      // (whole workgroup: the flag is uniform) tested once the first round of loads is back, see above
    const SegPre pre = seg_pre<OP>(out, src, threadIdx.x == 0 ? bdst : -1, baux, ex);
    double acc = 0.0;
    for (int32_t base = e0; base < e1; base += SegPlanHost::BLK_PASS)
      acc += seg_gather<OP, BLK_PER_THREAD, false>(p, src, ex, impl, [&](int x) {
        const int32_t e = base + (int32_t)threadIdx.x + 1024 * x;
        return e < e1 ? e : -1;
      });
    acc = wave_sum(acc);
    if (lane == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < SEG_WAVES; w++) tot += sh[w];     // fixed order
      seg_store<OP>(out, bdst, tot, pre, ex);
    }
    return;
  }
  const int task = ((int)blockIdx.x - p.B) * SEG_WAVES + (threadIdx.x >> 6);
  if (task < p.G) {
    const int32_t dst = p.grp_dst[task * 64 + lane];
    const int32_t aux = p.grp_aux[task * 64 + lane];
    const int32_t c0 = p.grp_off[task], c1 = p.grp_off[task + 1];
    if (skip) return;
    const SegPre pre = seg_pre<OP>(out, src, dst, aux, ex);
    double acc = 0.0;
    for (int32_t col = c0; col < c1; col += 8)       // a whole ELL group in one round: rows have <= 8 entries
      acc += seg_gather<OP, 8, true>(p, src, ex, impl, [&](int x) { return col + x < c1 ? (col + x) * 64 + lane : -1; });
    if (dst >= 0) seg_store<OP>(out, dst, acc, pre, ex);
  } else if (task < p.G + p.S) {
    const int sidx = task - p.G;
    const int32_t e0 = p.seg_beg[sidx], e1 = p.seg_end[sidx];
    const int32_t sdst = p.seg_dst[sidx];
    const int32_t saux = p.seg_aux[sidx];
    if (skip) return;
    const SegPre pre = seg_pre<OP>(out, src, lane == 0 ? sdst : -1, saux, ex);
    // up to 256 entries: four per lane; up to 1024: sixteen per lane - either way ONE round of index loads and ONE
    // round of gathers, all in flight together
    double acc;
    if (e1 - e0 <= 256)
      acc = seg_gather<OP, 4, false>(p, src, ex, impl, [&](int x) { const int32_t e = e0 + lane + 64 * x; return e < e1 ? e : -1; });
    else
      acc = seg_gather<OP, 16, false>(p, src, ex, impl, [&](int x) { const int32_t e = e0 + lane + 64 * x; return e < e1 ? e : -1; });
    acc = wave_sum(acc);
    if (lane == 0) seg_store<OP>(out, sdst, acc, pre, ex);
  }
}

void launch_segsum(const SegPlanView& p, SegOp op, const double* src, double* out, const SegExtra& ex, hipStream_t s) {
  const int tasks = p.G + p.S;
  if (tasks + p.B == 0) return;
#define KIN_SEG_LAUNCH(OPX)                                                                                              \
  if (p.B > 0) hipLaunchKernelGGL((segsum_kernel<OPX, 1024>), dim3((unsigned)(p.B + ceil_div(tasks, 16))), dim3(1024), 0, s, p, src, out, ex); \
  else hipLaunchKernelGGL((segsum_kernel<OPX, 256>), dim3((unsigned)ceil_div(tasks, 4)), dim3(256), 0, s, p, src, out, ex);
  switch (op) {
    case SEG_COEF_SET: KIN_SEG_LAUNCH(SEG_COEF_SET) break;
    case SEG_PROD_SUB: KIN_SEG_LAUNCH(SEG_PROD_SUB) break;
    case SEG_COEF_BDF: KIN_SEG_LAUNCH(SEG_COEF_BDF) break;
    case SEG_PROD_SUB_DIV: KIN_SEG_LAUNCH(SEG_PROD_SUB_DIV) break;
    case SEG_PROD_AUXSUB: KIN_SEG_LAUNCH(SEG_PROD_AUXSUB) break;
    case SEG_PROD_SET: KIN_SEG_LAUNCH(SEG_PROD_SET) break;
    case SEG_PROD_NEG: KIN_SEG_LAUNCH(SEG_PROD_NEG) break;
  }
#undef KIN_SEG_LAUNCH
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

// The same two kernels with the rate constants formed on the spot from a temperature (continuous-rate solves: the first
// reader of k after a change of T evaluates the Arrhenius law itself and stores k for the readers behind it - no launch of
// its own for the rate constants; calculator.jl:223-232 literally, as arrhenius_kernel)
__global__ __launch_bounds__(256) void rates_T_kernel(int R, ArrheniusAt at, double* __restrict__ k, const double* __restrict__ u,
                                                      const int32_t* __restrict__ x0, const int32_t* __restrict__ x1,
                                                      double* __restrict__ rate) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int32_t a = x0[r], b = x1[r];
  const double kk = arrhenius_one(at.Ea[r], at.A[r], 8.314462618 * at.T, at.has_kmax, at.k_max, at.t_mult);
  k[r] = kk;
  const double ub = b >= 0 ? u[b] : 1.0;
  rate[r] = kk * u[a] * ub;
}
__global__ __launch_bounds__(256) void drates_T_kernel(int R, ArrheniusAt at, double* __restrict__ k, const double* __restrict__ u,
                                                       const int32_t* __restrict__ x0, const int32_t* __restrict__ x1,
                                                       double* __restrict__ dr) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int32_t a = x0[r], b = x1[r];
  const double kk = arrhenius_one(at.Ea[r], at.A[r], 8.314462618 * at.T, at.has_kmax, at.k_max, at.t_mult);
  k[r] = kk;
  double d0, d1 = 0.0;
  if (b < 0) d0 = kk;
  else if (b == a) d0 = 2.0 * kk * u[a];
  else { d0 = kk * u[b]; d1 = kk * u[a]; }
  reinterpret_cast<double2*>(dr)[r] = make_double2(d0, d1);
}
void launch_rates_T(int64_t R, const ArrheniusAt& at, double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(rates_T_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s, (int)R, at, k, u, x0, x1, rate);
  KIN_HIP(hipGetLastError());
}
void launch_drates_T(int64_t R, const ArrheniusAt& at, double* k, const double* u, const int32_t* x0, const int32_t* x1, double* dr, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(drates_T_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s, (int)R, at, k, u, x0, x1, dr);
  KIN_HIP(hipGetLastError());
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
__global__ __launch_bounds__(256) void arrhenius_kernel(int n, const double* __restrict__ Ea, const double* __restrict__ A,
                                                        int has_kmax, double k_max, double t_mult, double T,
                                                        double* __restrict__ k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  k[i] = arrhenius_one(Ea[i], A[i], 8.314462618 * T, has_kmax, k_max, t_mult);
}

// (table-driven exp and the division-free Arrhenius of the table kernels: exp_tab.hpp)
// table[s][r]; one thread produces two consecutive reactions (16-byte stores), grid.y walks
// time stops so that each workgroup keeps its (Ea, A N_A t_mult) pairs in registers across
// TABLE_ROWS_PER_BLOCK rows; the rows' R T and 1 / (R T) are computed once per workgroup.
constexpr int TABLE_ROWS_PER_BLOCK = 32;   // 2: 1.91 ms, 8: 1.60 ms, 32: 1.54 ms at 14001 x 50000 (host call included)
__global__ __launch_bounds__(256) void rate_table_kernel(int n, int n_stops, const double* __restrict__ Ea,
                                                         const double* __restrict__ A, int has_kmax, double k_max,
                                                         double t_mult, const double* __restrict__ T,
                                                         double* __restrict__ table) {
  __shared__ double rt_s[TABLE_ROWS_PER_BLOCK], irt_s[TABLE_ROWS_PER_BLOCK];
  __shared__ double tab_s[512];
  tab_s[threadIdx.x] = kExp2Tab[threadIdx.x];
  tab_s[threadIdx.x + 256] = kExp2Tab[threadIdx.x + 256];
  const int s0 = blockIdx.y * TABLE_ROWS_PER_BLOCK;
  const int s1 = min(n_stops, s0 + TABLE_ROWS_PER_BLOCK);
  if ((int)threadIdx.x < s1 - s0) {
    const double RT = 8.314462618 * T[s0 + threadIdx.x];
    rt_s[threadIdx.x] = RT;
    irt_s[threadIdx.x] = 1.0 / RT;
  }
  __syncthreads();
  const int r = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (r >= n) return;
  const bool pair = (r + 1 < n);
  const double e0 = Ea[r], c0 = A[r] * 6.02214076e23 * t_mult, ic0 = 1.0 / c0;
  const double e1 = pair ? Ea[r + 1] : 0.0, c1 = pair ? A[r + 1] * 6.02214076e23 * t_mult : 1.0, ic1 = 1.0 / c1;
  const double inv_kmax = 1.0 / k_max;
  for (int s = s0; s < s1; s++) {
    const double RT = rt_s[s - s0], inv_RT = irt_s[s - s0];
    const double v0 = arrhenius_fast(e0, c0, ic0, RT, inv_RT, has_kmax, inv_kmax, tab_s);
    double* row = table + (size_t)s * n;
    if (pair && ((n & 1) == 0)) {
      const double v1 = arrhenius_fast(e1, c1, ic1, RT, inv_RT, has_kmax, inv_kmax, tab_s);
      *reinterpret_cast<double2*>(row + r) = make_double2(v0, v1);
    } else {
      row[r] = v0;
      if (pair) row[r + 1] = arrhenius_fast(e1, c1, ic1, RT, inv_RT, has_kmax, inv_kmax, tab_s);
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
// batched RHS sweep: B independent states, state-major layouts u[b][N], k[b][R], du[b][N]
// (the reference's own layout: sol.u is a Vector of Vectors).
//
// One 1024-thread workgroup owns one state at a time. The state's concentrations and its rate
// accumulator live in LDS (2 x 8N bytes: 160 kB at N = 10k, the whole LDS of a CU), so HBM
// sees exactly the algorithmic traffic: k[b][:] streamed once (coalesced 8 B/lane), u[b][:]
// read once, du[b][:] written once. The packed reaction records (16 B: six 16-bit species
// slots + four signed stoichiometric bytes) are shared by every state and stay L2 resident.
// Species rates are accumulated with FP64 LDS atomics (ds_add_f64). For N too large for LDS
// the kernel makes several passes over species tiles (du tile in LDS, u read through L1/L2).
// ------------------------------------------------------------------------------------------
struct SweepRec { uint32_t s01, s23; int32_t coef; uint32_t ops; };   // 0xFFFF = empty slot

// net rate of one record and its per-slot species / coefficients
template <bool U_IN_LDS>
__device__ __forceinline__ double sweep_net(const SweepRec& q, double kf, double kr, const double* u_s,
                                            const double* __restrict__ ub, uint32_t sl[4], int cf[4]) {
  sl[0] = q.s01 & 0xffffu; sl[1] = q.s01 >> 16; sl[2] = q.s23 & 0xffffu; sl[3] = q.s23 >> 16;
#pragma unroll
  for (int j = 0; j < 4; j++) cf[j] = (int)(int8_t)((uint32_t)q.coef >> (8 * j));
  if (q.ops == 0xffffffffu) {
    double uf = 1.0, ur = 1.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (sl[j] != 0xffffu) {
        const double v = U_IN_LDS ? u_s[sl[j]] : ub[sl[j]];
        const double v2 = (cf[j] == 2 || cf[j] == -2) ? v * v : v;
        if (cf[j] < 0) uf *= v2; else ur *= v2;
      }
    }
    return kf * uf - kr * ur;
  }
  // explicit operands (a species on both sides of the reaction)
  const uint32_t a = q.ops & 0xffffu, c = q.ops >> 16;
  double net = kf * (U_IN_LDS ? u_s[a] : ub[a]);
  if (c != 0xffffu) net *= (U_IN_LDS ? u_s[c] : ub[c]);
  return net;
}

// Each thread handles SWEEP_ILP records per trip and issues all of their global loads before
// the first use: with one 1024-thread workgroup per CU (LDS bound) this is what keeps enough
// bytes in flight to cover HBM latency. ADJ: record p pairs reactions (2p, 2p+1), so the rate
// constants stream as one coalesced double2 per lane and do not depend on any index load.
constexpr int SWEEP_ILP = 8;

template <bool U_IN_LDS, bool ADJ>
__global__ __launch_bounds__(1024) void sweep_lds_kernel(int N, int R, int P, int B, int tile, int n_tiles,
                                                         const SweepRec* __restrict__ rec, const int2* __restrict__ pair_k,
                                                         const double* __restrict__ u, const double* __restrict__ k_b,
                                                         const double* __restrict__ k_1, double* __restrict__ du) {
  extern __shared__ double lds[];
  double* du_s = lds;
  double* u_s = lds + tile;
  const int tid = threadIdx.x;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const double* ub = u + (size_t)b * N;
    const double* kb = k_b ? k_b + (size_t)b * R : k_1;
    double* dub = du + (size_t)b * N;
    if (U_IN_LDS) {
      for (int i = tid * 2; i < N; i += 2048) {
        if (i + 1 < N) *reinterpret_cast<double2*>(u_s + i) = *reinterpret_cast<const double2*>(ub + i);
        else u_s[i] = ub[i];
      }
    }
    for (int t = 0; t < n_tiles; t++) {
      const int lo = t * tile, hi = min(N, lo + tile);
      for (int i = tid; i < hi - lo; i += 1024) du_s[i] = 0.0;
      __syncthreads();
      for (int p0 = tid; p0 < P; p0 += 1024 * SWEEP_ILP) {
        SweepRec q[SWEEP_ILP];
        double kf[SWEEP_ILP], kr[SWEEP_ILP];
        if (ADJ) {
#pragma unroll
          for (int x = 0; x < SWEEP_ILP; x++) {
            const int p = p0 + x * 1024;
            if (p < P) {
              const double2 kk = *reinterpret_cast<const double2*>(kb + 2 * (size_t)p);
              kf[x] = kk.x; kr[x] = kk.y;
              q[x] = rec[p];
            }
          }
        } else {
          int2 kk[SWEEP_ILP];
#pragma unroll
          for (int x = 0; x < SWEEP_ILP; x++) {
            const int p = p0 + x * 1024;
            if (p < P) { kk[x] = pair_k[p]; q[x] = rec[p]; }
          }
#pragma unroll
          for (int x = 0; x < SWEEP_ILP; x++) {
            const int p = p0 + x * 1024;
            if (p < P) { kf[x] = kb[kk[x].x]; kr[x] = kk[x].y >= 0 ? kb[kk[x].y] : 0.0; }
          }
        }
#pragma unroll
        for (int x = 0; x < SWEEP_ILP; x++) {
          const int p = p0 + x * 1024;
          if (p < P) {
            uint32_t sl[4]; int cf[4];
            const double net = sweep_net<U_IN_LDS>(q[x], kf[x], kr[x], u_s, ub, sl, cf);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int sp = (int)sl[j];
              if (sp != 0xffff && sp >= lo && sp < hi)
                __hip_atomic_fetch_add(du_s + (sp - lo), (double)cf[j] * net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
      __syncthreads();
      for (int i = tid * 2; i < hi - lo; i += 2048) {
        if (i + 1 < hi - lo && ((lo & 1) == 0)) *reinterpret_cast<double2*>(dub + lo + i) = *reinterpret_cast<double2*>(du_s + i);
        else { dub[lo + i] = du_s[i]; if (i + 1 < hi - lo) dub[lo + i + 1] = du_s[i + 1]; }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Register-resident variant (adjacent pairs, state fits LDS): the reaction records a thread works on
// are the same for every state, so they are loaded ONCE into registers and the only per-state global
// traffic left is the algorithmic one: k[b] streamed as double2, u[b] in, du[b] out.
// Record = four 14-bit species labels with fixed roles (fields 0, 1: reactant instances of the forward
// reaction, fields 2, 3: its product instances; network.cpp) - no coefficients, no side flags, no
// branches: SQ counters of the previous format (2-bit codes per slot, empty-slot tests) showed ~100
// VALU + ~50 SALU instructions per record and the VALU 47 % busy next to a 54 % busy LDS. Unused fields
// point at a per-lane dummy entry (u = 1, du discarded) behind the N real ones.
// ------------------------------------------------------------------------------------------
constexpr int SWEEP_DUMMY = 64;   // dummy entries behind u_s / du_s, one per lane
__device__ __forceinline__ void sweep_apply(uint2 w, double2 kk, const double* u_s, double* du_s) {
  const uint32_t l0 = w.x & 0x3fffu, l1 = (w.x >> 14) & 0x3fffu, l2 = (w.x >> 28) | ((w.y & 0x3ffu) << 4), l3 = (w.y >> 10) & 0x3fffu;
#if defined(KIN_SWEEP_PROBE) && KIN_SWEEP_PROBE == 1   // timing only (wrong results): no LDS operand reads
  const double uf = 1.0 + (double)(l0 + l1), ur = 1.0 + (double)(l2 + l3);
#else
  const double uf = u_s[l0] * u_s[l1], ur = u_s[l2] * u_s[l3];
#endif
  const double net = kk.x * uf - kk.y * ur;
#if defined(KIN_SWEEP_PROBE)                            // timing only: no LDS atomics, the net rate kept alive
  if (net == 12345.678) du_s[l0] = net;
  return;
#endif
  __hip_atomic_fetch_add(du_s + l0, -net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(du_s + l1, -net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(du_s + l2, net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(du_s + l3, net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// rate constants of pair p: adjacent layout = one double2 at k[2p]; block layout (BLK: all forward reactions
// first, their reverses in the same order behind them - what duplicate_reverse produces, cde.jl:299-309) =
// k[p] and k[P + p], two coalesced 8-byte streams
template <bool BLK>
__device__ __forceinline__ double2 load_kpair(const double* __restrict__ kb, int p, int P) {
  if (BLK) return make_double2(kb[p], kb[(size_t)P + p]);
  return *reinterpret_cast<const double2*>(kb + 2 * (size_t)p);
}

// TR = records per thread kept in registers (compile time, fully unrolled); records beyond
// TR*1024 are streamed as 8-byte words. ILP = records whose loads are issued together.
template <int TR, int ILP, bool BLK, int BS>
__global__ __launch_bounds__(BS) void sweep_reg_kernel(int N, int R, int P, int B, int tile,
                                                         const uint2* __restrict__ rec64, const int32_t* __restrict__ copy_species,
                                                         int n_copy, const double* __restrict__ u,
                                                         const double* __restrict__ k_b, const double* __restrict__ k_1,
                                                         double* __restrict__ du) {
  extern __shared__ double lds[];
  double* du_s = lds;
  double* u_s = lds + tile;
  const int tid = threadIdx.x;
  constexpr int UPT = 5;            // double2 per thread of the staged state (N <= BS0)
  // padding record of this lane: all four fields on the lane's dummy entry
  const uint64_t dl = (uint64_t)N + (uint64_t)(tid & 63);
  const uint64_t ew = dl | (dl << 14) | (dl << 28) | (dl << 42);
  const uint2 EMPTY = {(uint32_t)ew, (uint32_t)(ew >> 32)};
  if (tid < SWEEP_DUMMY) { u_s[N + tid] = 1.0; du_s[N + tid] = 0.0; }
  // split accumulators of the most referenced species (network.cpp): entry N + 64 + tid mirrors species csp
  const int csp = tid < n_copy ? copy_species[min(tid, max(n_copy - 1, 0))] : -1;
  if (csp >= 0) du_s[N + SWEEP_DUMMY + tid] = 0.0;
  // Every global load below is UNCONDITIONAL (indices clamped into valid memory; a record beyond the end is the lane's
  // padding record, whose rate constants meet dummy entries): a load behind a branch makes the compiler wait for
  // vmcnt(0) wherever it cannot count the loads in flight - before the state's barrier (draining the prefetched rate
  // constants), between the batches of the register-resident records and in the streamed loop (tiled_kernels.hip has
  // the same rule and the measurement behind it).
  const int Pm1 = P - 1, Bm1 = B - 1, csp_c = max(csp, 0);
  uint2 rc[TR > 0 ? TR : 1];
#pragma unroll
  for (int i = 0; i < TR; i++) {
    const int p = tid + i * BS;
    const uint2 w = rec64[min(p, Pm1)];
    rc[i].x = p < P ? w.x : EMPTY.x;
    rc[i].y = p < P ? w.y : EMPTY.y;
  }
  // software pipeline over states: the next state's u travels HBM -> registers while this state's
  // reactions are processed; du is written out and re-zeroed in one pass. N is even (host check).
  double2 un[UPT];
  int b = blockIdx.x;
  {
    const double* ub = u + (size_t)min(b, Bm1) * N;
#pragma unroll
    for (int x = 0; x < UPT; x++) {
      // (component-wise: a whole-vector assignment here leaves `un` in scratch memory - ROCm 7.2 clang)
      const double2 t = *reinterpret_cast<const double2*>(ub + min((tid + x * BS) * 2, N - 2));
      un[x].x = t.x; un[x].y = t.y;
    }
  }
  for (int i = tid * 2; i < N; i += (2 * BS)) *reinterpret_cast<double2*>(du_s + i) = make_double2(0.0, 0.0);
  double ucn = u[(size_t)min(b, Bm1) * N + csp_c];
  // the first batch of rate constants of a state is requested before the previous state's barrier /
  // write-out / staging, so the k stream does not drain at state boundaries
  constexpr bool KPRE = TR >= ILP;
  double2 k0[KPRE ? ILP : 1];
  if (KPRE) {
    const double* kb = k_b ? k_b + (size_t)min(b, Bm1) * R : k_1;
#pragma unroll
    for (int x = 0; x < ILP; x++) k0[x] = load_kpair<BLK>(kb, min(tid + x * BS, Pm1), P);
  }
  for (; b < B; b += gridDim.x) {
    const double* kb = k_b ? k_b + (size_t)b * R : k_1;
    double* dub = du + (size_t)b * N;
#pragma unroll
    for (int x = 0; x < UPT; x++) {
      const int i = (tid + x * BS) * 2;
      if (i < N) *reinterpret_cast<double2*>(u_s + i) = un[x];
    }
    if (csp >= 0) u_s[N + SWEEP_DUMMY + tid] = ucn;
    __syncthreads();
    const int bn = b + gridDim.x;
    ucn = u[(size_t)min(bn, Bm1) * N + csp_c];
    {
      const double* ub = u + (size_t)min(bn, Bm1) * N;
#pragma unroll
      for (int x = 0; x < UPT; x++) {
        const double2 t = *reinterpret_cast<const double2*>(ub + min((tid + x * BS) * 2, N - 2));
        un[x].x = t.x; un[x].y = t.y;
      }
    }
    // register-resident records
#pragma unroll
    for (int i0 = 0; i0 < TR; i0 += ILP) {
      double2 kk[ILP];
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        if (KPRE && i0 == 0) kk[x] = k0[x];
        else kk[x] = load_kpair<BLK>(kb, min(tid + (i0 + x) * BS, Pm1), P);
      }
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        uint2 w = rc[i0 + x];
        // opaque to the optimiser: without this the loop-invariant decode (slot indices and lane
        // masks of every record) is hoisted out of the state loop and the kernel spills
        asm volatile("" : "+v"(w.x), "+v"(w.y));
        sweep_apply(w, kk[x], u_s, du_s);
      }
    }
    // streamed records
    for (int p0 = tid + TR * BS; p0 < P; p0 += BS * ILP) {
      double2 kk[ILP];
      uint2 w[ILP];
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        const int p = p0 + x * BS;
        const bool ok = p < P;
        kk[x] = load_kpair<BLK>(kb, min(p, Pm1), P);
        const uint2 q = rec64[min(p, Pm1)];
        w[x].x = ok ? q.x : EMPTY.x;
        w[x].y = ok ? q.y : EMPTY.y;
      }
#pragma unroll
      for (int x = 0; x < ILP; x++) sweep_apply(w[x], kk[x], u_s, du_s);
    }
    if (KPRE) {
      const double* kn = k_b ? k_b + (size_t)min(bn, Bm1) * R : k_1;
#pragma unroll
      for (int x = 0; x < ILP; x++) k0[x] = load_kpair<BLK>(kn, min(tid + x * BS, Pm1), P);
    }
    __syncthreads();
    if (n_copy > 0) {   // fold the split accumulators back into their species
      if (csp >= 0) {
        const double v = du_s[N + SWEEP_DUMMY + tid];
        du_s[N + SWEEP_DUMMY + tid] = 0.0;
        __hip_atomic_fetch_add(du_s + csp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
    }
    for (int i = tid * 2; i < N; i += (2 * BS)) {
      *reinterpret_cast<double2*>(dub + i) = *reinterpret_cast<double2*>(du_s + i);
      *reinterpret_cast<double2*>(du_s + i) = make_double2(0.0, 0.0);
    }
    // no barrier needed here: the next trip only touches u_s before its own barrier
  }
}

template <int TR, int ILP, bool BLK, int BS>
static void launch_sweep_reg_t(int grid, size_t smem, int N, int R, int P, int B, int tile, const void* rec64,
                               const int32_t* copy_species, int n_copy, const double* u, const double* k_b, const double* k_1,
                               double* du, hipStream_t s) {
  // per launch, not cached: the attribute belongs to the (function, device) pair and costs ~1 us
  KIN_HIP(hipFuncSetAttribute((const void*)sweep_reg_kernel<TR, ILP, BLK, BS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((sweep_reg_kernel<TR, ILP, BLK, BS>), dim3(grid), dim3(BS), smem, s, N, R, P, B, tile, (const uint2*)rec64,
                     copy_species, n_copy, u, k_b, k_1, du);
}

// ------------------------------------------------------------------------------------------
// General variant for a state that fits LDS but whose reactions are NOT all paired with their reverses in
// one of the two regular layouts (what the low-k cutoff leaves behind, or hand-written networks): the same
// fixed-role 8-byte label records (16-bit labels, per-lane dummies), one per forward/reverse pair or single
// reaction, plus the (kf, kr) reaction indices of the record (kr = -1: no reverse); the rate constants are
// gathered by index (near-coalesced, the records follow the reaction order). Reactions with a species on both
// sides are all-dummy in the stream and take the slow path with the 16-byte record. Replaces the coefficient-
// decoding loop of sweep_lds_kernel for this case (C3 CRN minus a random 30 % of its reactions, B = 4096:
// 0.66 ms -> 0.52 ms, tools/unpaired_sweep.py).
// ------------------------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(BS) void sweep_gen_kernel(int N, int R, int P, int B, int tile, const uint2* __restrict__ rec8,
                                                         const int2* __restrict__ pair_k, const SweepRec* __restrict__ rec,
                                                         const int32_t* __restrict__ expl, int n_expl,
                                                         const double* __restrict__ u, const double* __restrict__ k_b,
                                                         const double* __restrict__ k_1, double* __restrict__ du) {
  extern __shared__ double lds[];
  double* du_s = lds;
  double* u_s = lds + tile;
  const int tid = threadIdx.x;
  constexpr int ILP = 4;
  const uint32_t dl = (uint32_t)(N + (tid & 63));
  const uint2 EMPTY = {dl | (dl << 16), dl | (dl << 16)};
  if (tid < SWEEP_DUMMY) { u_s[N + tid] = 1.0; du_s[N + tid] = 0.0; }
  for (int i = tid; i < N; i += BS) du_s[i] = 0.0;
  // the next state's u travels HBM -> registers while this state's records are processed (N <= 10176)
  constexpr int UPT = 10;
  double un[UPT];
  int b = blockIdx.x;
#pragma unroll
  for (int x = 0; x < UPT; x++) { const int i = tid + x * BS; un[x] = (b < B && i < N) ? u[(size_t)b * N + i] : 0.0; }
  for (; b < B; b += gridDim.x) {
    const double* kb = k_b ? k_b + (size_t)b * R : k_1;
    double* dub = du + (size_t)b * N;
#pragma unroll
    for (int x = 0; x < UPT; x++) { const int i = tid + x * BS; if (i < N) u_s[i] = un[x]; }
    __syncthreads();
    const int bn = b + gridDim.x;
#pragma unroll
    for (int x = 0; x < UPT; x++) { const int i = tid + x * BS; un[x] = (bn < B && i < N) ? u[(size_t)bn * N + i] : 0.0; }
    for (int qq = tid; qq < P; qq += BS * ILP) {
      uint2 w[ILP];
      int2 ki[ILP];
      double kf[ILP], kr[ILP];
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        const int p = qq + x * BS;
        w[x] = p < P ? rec8[p] : EMPTY;
        ki[x] = p < P ? pair_k[p] : make_int2(-1, -1);
      }
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        kf[x] = ki[x].x >= 0 ? kb[ki[x].x] : 0.0;
        kr[x] = ki[x].y >= 0 ? kb[ki[x].y] : 0.0;
      }
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        const uint32_t l0 = w[x].x & 0xffffu, l1 = w[x].x >> 16, l2 = w[x].y & 0xffffu, l3 = w[x].y >> 16;
#if defined(KIN_SWEEP_PROBE)   // timing only (wrong results): =1 no LDS work at all, =2 no LDS atomics
        const double net = KIN_SWEEP_PROBE == 1 ? kf[x] * (1.0 + (double)(l0 + l1)) - kr[x] * (1.0 + (double)(l2 + l3))
                                                : kf[x] * (u_s[l0] * u_s[l1]) - kr[x] * (u_s[l2] * u_s[l3]);
        if (net == 12345.678) du_s[l0] = net;
        continue;
#else
        const double net = kf[x] * (u_s[l0] * u_s[l1]) - kr[x] * (u_s[l2] * u_s[l3]);
#endif
        __hip_atomic_fetch_add(du_s + l0, -net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(du_s + l1, -net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(du_s + l2, net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(du_s + l3, net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    // reactions with a species on both sides: net = kf u[a] u[c], coefficients from the 16-byte record
    for (int i = tid; i < n_expl; i += BS) {
      const int p = expl[i];
      const SweepRec q = rec[p];
      const uint32_t a = q.ops & 0xffffu, c = q.ops >> 16;
      double net = kb[pair_k[p].x] * u_s[a];
      if (c != 0xffffu) net *= u_s[c];
      const uint32_t sl[4] = {q.s01 & 0xffffu, q.s01 >> 16, q.s23 & 0xffffu, q.s23 >> 16};
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (sl[j] != 0xffffu)
          __hip_atomic_fetch_add(du_s + sl[j], (double)(int)(int8_t)((uint32_t)q.coef >> (8 * j)) * net, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    for (int i = tid; i < N; i += BS) { dub[i] = du_s[i]; du_s[i] = 0.0; }
    // no barrier needed here: the next trip only touches u_s before its own barrier
  }
}

// ------------------------------------------------------------------------------------------
// Large-N sweep (2 x 8N bytes exceed the LDS; C5: N = 50k). Species are relabelled by popularity:
// the H most referenced species ("hubs", ~81 % of all slot references under the Zipf wiring) keep
// u and du in LDS exactly as above. Per state, one workgroup
//   1. gathers u into label order: hubs -> LDS, tail -> its private global scratch row `ut`
//      (written and read by the same CU: served by L2 / Infinity Cache, not HBM);
//   2. streams the records and the state's k (coalesced double2) once, accumulates the hubs' du with
//      LDS atomics and stores every record's net rate to its private `netbuf` row (coalesced 8 B/lane).
//      Records are 8 bytes: four 16-bit labels with fixed roles (fields 0, 1 reactant instances of the
//      forward reaction, fields 2, 3 its product instances) - no coefficient decoding, as in
//      sweep_reg_kernel; labels [H, H + 64) are per-lane dummy entries in LDS (u = 1, du discarded),
//      tail labels start at H + 64. Reactions with a species on both sides (rare) take a slow path;
//   3. writes the hubs' du out, then re-uses the whole LDS as accumulator for the tail, tile by tile:
//      a precomputed list of (record, local label, coefficient) entries per tail tile, sorted by
//      record, gathers coef * netbuf[record] into LDS; the tile is then written out.
// (FP64 L2 atomics for the tail were measured first: 80 M global_atomic_add_f64 per launch cost
// 1.3 ms of 2.5 ms.) HBM traffic per state is the algorithmic one: k[b][:] + u[b][:] in, du[b][:] out.
// ------------------------------------------------------------------------------------------
template <bool ADJ>
__global__ __launch_bounds__(1024) void sweep_big_kernel(int N, int R, int P, int B, int H, int n_tail_tiles,
                                                         const uint2* __restrict__ rec8, const SweepRec* __restrict__ rec,
                                                         const int32_t* __restrict__ expl, int n_expl,
                                                         const int2* __restrict__ pair_k,
                                                         const int32_t* __restrict__ spec_of_label,
                                                         const int32_t* __restrict__ tail_ptr, const uint2* __restrict__ tail_ent,
                                                         double* __restrict__ scratch, const double* __restrict__ u,
                                                         const double* __restrict__ k_b, const double* __restrict__ k_1,
                                                         double* __restrict__ du, int by_species) {
  extern __shared__ double lds[];
  const int HL = H + SWEEP_DUMMY;            // LDS entries per array: hubs + per-lane dummies
  double* du_s = lds;
  double* u_s = lds + HL;
  const int tid = threadIdx.x;
  const int NT = N - H, TT = 2 * H;
  double* ut = scratch + (size_t)blockIdx.x * ((size_t)NT + P);
  double* netbuf = ut + NT;
  constexpr int ILP = 6, PERM_ILP = 8;   // 6 records per thread in flight: 1.46 ms at C5 (4: 1.52, 8: 2.26)
  const uint32_t dl = (uint32_t)(H + (tid & 63));
  const uint2 EMPTY = {dl | (dl << 16), dl | (dl << 16)};
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const double* ub = u + (size_t)b * N;
    const double* kb = k_b ? k_b + (size_t)b * R : k_1;
    double* dub = du + (size_t)b * N;
    // 1. permuting gather, PERM_ILP independent element chains per thread in flight. With tail operands addressed by
    // species id (by_species) only the hubs are gathered: the record stream reads tail operands straight from u[b].
    const double* tsrc = by_species ? ub : ut;
    for (int l0 = tid; l0 < (by_species ? H : N); l0 += 1024 * PERM_ILP) {
      int sp[PERM_ILP];
      double v[PERM_ILP];
#pragma unroll
      for (int x = 0; x < PERM_ILP; x++) { const int l = l0 + x * 1024; sp[x] = l < N ? spec_of_label[l] : -1; }
#pragma unroll
      for (int x = 0; x < PERM_ILP; x++) v[x] = sp[x] >= 0 ? ub[sp[x]] : 0.0;
#pragma unroll
      for (int x = 0; x < PERM_ILP; x++) {
        const int l = l0 + x * 1024;
        if (l < H) { u_s[l] = v[x]; du_s[l] = 0.0; }
        else if (l < N) ut[l - H] = v[x];
      }
    }
    if (tid < SWEEP_DUMMY) { u_s[H + tid] = 1.0; du_s[H + tid] = 0.0; }
    __syncthreads();   // workgroup-scope release/acquire: the scratch row is visible to every wave of this CU
    // 2. record stream
    for (int qq = tid; qq < P; qq += 1024 * ILP) {
      uint2 w[ILP];
      double kf[ILP], kr[ILP];
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        const int p = qq + x * 1024;
        if (p < P) {
          w[x] = rec8[p];
          if (ADJ) {
            const double2 kk = *reinterpret_cast<const double2*>(kb + 2 * (size_t)p);
            kf[x] = kk.x; kr[x] = kk.y;
          } else {
            const int2 kk = pair_k[p];
            kf[x] = kb[kk.x]; kr[x] = kk.y >= 0 ? kb[kk.y] : 0.0;
          }
        } else {
          w[x] = EMPTY;
          kf[x] = kr[x] = 0.0;
        }
      }
      // all operand loads of the ILP records are issued before the first product (the labels are cheap to
      // re-extract, so only the operand values stay live across the two phases)
      double uv[ILP][4];
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        const uint32_t sl[4] = {w[x].x & 0xffffu, w[x].x >> 16, w[x].y & 0xffffu, w[x].y >> 16};
#pragma unroll
        for (int j = 0; j < 4; j++) uv[x][j] = (int)sl[j] < HL ? u_s[sl[j]] : tsrc[(int)sl[j] - HL];
      }
#pragma unroll
      for (int x = 0; x < ILP; x++) {
        const double net = kf[x] * (uv[x][0] * uv[x][1]) - kr[x] * (uv[x][2] * uv[x][3]);
        const int p = qq + x * 1024;
        if (p < P) netbuf[p] = net;
        const uint32_t sl[4] = {w[x].x & 0xffffu, w[x].x >> 16, w[x].y & 0xffffu, w[x].y >> 16};
#pragma unroll
        for (int j = 0; j < 4; j++)
          if ((int)sl[j] < HL)
            __hip_atomic_fetch_add(du_s + sl[j], j < 2 ? -net : net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    // 2b. reactions with a species on both sides (all-dummy in rec8): net = kf u[a] u[c], coefficients from
    // the 16-byte record; operands gathered straight from the state
    if (n_expl > 0) {
      __syncthreads();   // their netbuf entries were written (with a meaningless value) by the stream above
      for (int i = tid; i < n_expl; i += 1024) {
        const int p = expl[i];
        const SweepRec q = rec[p];
        const double kf = ADJ ? kb[2 * (size_t)p] : kb[pair_k[p].x];
        const uint32_t a = q.ops & 0xffffu, c = q.ops >> 16;
        // (16-byte records carry labels: tail operands by label, through the label -> species table when there is no `ut`)
        double net = kf * ((int)a < H ? u_s[a] : (by_species ? ub[spec_of_label[a]] : ut[(int)a - H]));
        if (c != 0xffffu) net *= ((int)c < H ? u_s[c] : (by_species ? ub[spec_of_label[c]] : ut[(int)c - H]));
        netbuf[p] = net;
        const uint32_t sl[4] = {q.s01 & 0xffffu, q.s01 >> 16, q.s23 & 0xffffu, q.s23 >> 16};
#pragma unroll
        for (int j = 0; j < 4; j++)
          if ((int)sl[j] < H)
            __hip_atomic_fetch_add(du_s + sl[j], (double)(int)(int8_t)((uint32_t)q.coef >> (8 * j)) * net, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    // 3a. hubs out
    for (int l0 = tid; l0 < H; l0 += 1024 * PERM_ILP) {
      int sp[PERM_ILP];
#pragma unroll
      for (int x = 0; x < PERM_ILP; x++) { const int l = l0 + x * 1024; sp[x] = l < H ? spec_of_label[l] : -1; }
#pragma unroll
      for (int x = 0; x < PERM_ILP; x++) if (sp[x] >= 0) dub[sp[x]] = du_s[l0 + x * 1024];
    }
    // 3b. tail tiles: the whole LDS (u_s is dead now) accumulates TT = 2H labels per pass
    for (int t = 0; t < n_tail_tiles; t++) {
      __syncthreads();
      for (int i = tid; i < TT; i += 1024) lds[i] = 0.0;
      __syncthreads();
      const int e0 = tail_ptr[t], e1 = tail_ptr[t + 1];
      for (int ee = e0 + tid; ee < e1; ee += 1024 * PERM_ILP) {
        uint2 en[PERM_ILP];
        double nv[PERM_ILP];
#pragma unroll
        for (int x = 0; x < PERM_ILP; x++) { const int e = ee + x * 1024; en[x] = e < e1 ? tail_ent[e] : make_uint2(0xffffffffu, 0u); }
#pragma unroll
        for (int x = 0; x < PERM_ILP; x++) nv[x] = en[x].x != 0xffffffffu ? netbuf[en[x].x] : 0.0;
#pragma unroll
        for (int x = 0; x < PERM_ILP; x++)
          if (en[x].x != 0xffffffffu)
            __hip_atomic_fetch_add(lds + (en[x].y & 0xffffffu), (double)((int)en[x].y >> 24) * nv[x], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
      const int base = H + t * TT, cnt = min(TT, N - base);
      for (int l0 = tid; l0 < cnt; l0 += 1024 * PERM_ILP) {
        int sp[PERM_ILP];
#pragma unroll
        for (int x = 0; x < PERM_ILP; x++) { const int l = l0 + x * 1024; sp[x] = l < cnt ? spec_of_label[base + l] : -1; }
#pragma unroll
        for (int x = 0; x < PERM_ILP; x++) if (sp[x] >= 0) dub[sp[x]] = lds[l0 + x * 1024];
      }
    }
    __syncthreads();
  }
}

template <bool ADJ>
static void launch_sweep_big_t(int grid, int N, int R, int P, int B, int H, int n_tail_tiles, const void* rec8, const void* rec,
                               const int32_t* expl, int n_expl, const void* pair_k, const int32_t* spec_of_label,
                               const int32_t* tail_ptr, const void* tail_ent, double* scratch, const double* u, const double* k_b,
                               const double* k_1, double* du, int by_species, hipStream_t s) {
  // per launch, not cached: the attribute belongs to the (function, device) pair and costs ~1 us
  KIN_HIP(hipFuncSetAttribute((const void*)sweep_big_kernel<ADJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((sweep_big_kernel<ADJ>), dim3(grid), dim3(1024), (size_t)2 * (H + SWEEP_DUMMY) * 8, s, N, R, P, B, H,
                     n_tail_tiles, (const uint2*)rec8, (const SweepRec*)rec, expl, n_expl, (const int2*)pair_k, spec_of_label,
                     tail_ptr, (const uint2*)tail_ent, scratch, u, k_b, k_1, du, by_species);
}

// `scratch` holds min(B, n_cu) rows of (N - H) + P doubles; n_cu = compute units of the handle's device
void launch_sweep_big(int n_cu, int64_t N, int64_t R, int64_t P, int64_t B, bool adjacent, int32_t H, int32_t n_tail_tiles, const void* rec8,
                      const void* rec, const int32_t* expl, int32_t n_expl, const void* pair_k, const int32_t* spec_of_label,
                      const int32_t* tail_ptr, const void* tail_ent, double* scratch, const double* u, const double* k_b,
                      const double* k_1, double* du, bool tail_by_species, hipStream_t s) {
  if (B == 0) return;
  const int grid = (int)std::min<int64_t>(B, n_cu);
  const bool adj = adjacent && ((((uintptr_t)(k_b ? k_b : k_1)) & 15) == 0);
  if (adj) launch_sweep_big_t<true>(grid, (int)N, (int)R, (int)P, (int)B, H, n_tail_tiles, rec8, rec, expl, n_expl, pair_k, spec_of_label, tail_ptr, tail_ent, scratch, u, k_b, k_1, du, tail_by_species ? 1 : 0, s);
  else launch_sweep_big_t<false>(grid, (int)N, (int)R, (int)P, (int)B, H, n_tail_tiles, rec8, rec, expl, n_expl, pair_k, spec_of_label, tail_ptr, tail_ent, scratch, u, k_b, k_1, du, tail_by_species ? 1 : 0, s);
  KIN_HIP(hipGetLastError());
}

template <bool U_IN_LDS, bool ADJ>
static void launch_sweep_t(int grid, size_t smem, int N, int R, int P, int B, int tile, int n_tiles, const void* rec,
                           const void* pair_k, const double* u, const double* k_b, const double* k_1, double* du, hipStream_t s) {
  // per launch, not cached: the attribute belongs to the (function, device) pair and costs ~1 us
  KIN_HIP(hipFuncSetAttribute((const void*)sweep_lds_kernel<U_IN_LDS, ADJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((sweep_lds_kernel<U_IN_LDS, ADJ>), dim3(grid), dim3(1024), smem, s, N, R, P, B, tile, n_tiles,
                     (const SweepRec*)rec, (const int2*)pair_k, u, k_b, k_1, du);
}

void launch_sweep(int n_cu, int64_t N, int64_t R, int64_t P, int64_t B, bool adjacent, bool block, const void* rec, const void* pair_k,
                  const void* rec64, const int32_t* copy_species, int n_copy, const void* gen_rec8, const int32_t* gen_expl,
                  int n_gen_expl, const double* u, const double* k_b, const double* k_1, double* du, hipStream_t s) {
  if (B == 0) return;
  const size_t lds_max = 160 * 1024;
  const int grid = (int)std::min<int64_t>(B, n_cu);
  // the double2 fast path also needs 16-byte aligned rows of k: R even (checked by the host) and an aligned base
  const bool adj = adjacent && ((((uintptr_t)(k_b ? k_b : k_1)) & 15) == 0);
  if ((size_t)(2 * N) * 8 <= lds_max) {
    const int tile = (int)((N + 1) / 2 * 2);
    const size_t smem = (size_t)(tile + N) * 8;
    constexpr int use_reg = 8;
    // register-resident path: both LDS arrays carry SWEEP_DUMMY extra entries (per-lane dummy species)
    // register-resident path: reactions paired as (2p, 2p+1) [adjacent] or (p, P+p) [block]
    if ((adj || block) && rec64 && use_reg >= 0 && ((((uintptr_t)u) | ((uintptr_t)du)) & 15) == 0 && N % 2 == 0 &&
        (size_t)(2 * (N + SWEEP_DUMMY + n_copy)) * 8 <= lds_max && n_copy <= 1024) {
      const int rtile = (int)N + SWEEP_DUMMY + n_copy + (n_copy & 1);   // even: keeps u_s 16-byte aligned
      const size_t rsmem = (size_t)2 * rtile * 8;
      // smaller states run in smaller workgroups, several per CU (LDS permitting): more states in flight per CU and
      // enough records per thread to keep them register-resident (C2, N = 1000: 0.073 -> 0.050 ms)
      const int bs = (N <= 2560 && n_copy <= 256) ? 256 : ((N <= 5120 && n_copy <= 512) ? 512 : 1024);
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / bs, lds_max / rsmem));
      const int rgrid = (int)std::min<int64_t>(B, (int64_t)n_cu * per_cu);
      const int64_t Tr = P / bs;   // full record rows available for residency
#define KIN_REG_ARGS rgrid, rsmem, (int)N, (int)R, (int)P, (int)B, rtile, rec64, copy_species, n_copy, u, k_b, k_1, du, s
#define KIN_REG_GO(TT, II)                                                                   \
  do {                                                                                       \
    if (bs == 256) { if (adj) launch_sweep_reg_t<TT, II, false, 256>(KIN_REG_ARGS); else launch_sweep_reg_t<TT, II, true, 256>(KIN_REG_ARGS); }   \
    else if (bs == 512) { if (adj) launch_sweep_reg_t<TT, II, false, 512>(KIN_REG_ARGS); else launch_sweep_reg_t<TT, II, true, 512>(KIN_REG_ARGS); } \
    else { if (adj) launch_sweep_reg_t<TT, II, false, 1024>(KIN_REG_ARGS); else launch_sweep_reg_t<TT, II, true, 1024>(KIN_REG_ARGS); }       \
  } while (0)
      const int want = (int)std::min<int64_t>(use_reg, Tr);
      if (want >= 16) KIN_REG_GO(16, 4);
      else if (want >= 12) KIN_REG_GO(12, 4);
      else if (want >= 8) KIN_REG_GO(8, 4);
      else if (want >= 4) KIN_REG_GO(4, 4);
      else if (use_reg == 1) KIN_REG_GO(0, 8);
      else KIN_REG_GO(0, 4);
#undef KIN_REG_GO
#undef KIN_REG_ARGS
      KIN_HIP(hipGetLastError());
      return;
    }
    if (gen_rec8 && (size_t)(2 * (N + SWEEP_DUMMY)) * 8 <= lds_max) {
      const int gtile = (int)N + SWEEP_DUMMY;
      const size_t gsmem = (size_t)2 * gtile * 8;
      const int bs = N <= 2560 ? 256 : (N <= 5120 ? 512 : 1024);     // smaller states: smaller workgroups, several per CU
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / bs, lds_max / gsmem));
      const int ggrid = (int)std::min<int64_t>(B, (int64_t)n_cu * per_cu);
#define KIN_GEN_GO(BSZ)                                                                                                                 \
  do {                                                                                                                                  \
    KIN_HIP(hipFuncSetAttribute((const void*)sweep_gen_kernel<BSZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
    hipLaunchKernelGGL(sweep_gen_kernel<BSZ>, dim3(ggrid), dim3(BSZ), gsmem, s, (int)N, (int)R, (int)P, (int)B, gtile,                  \
                       (const uint2*)gen_rec8, (const int2*)pair_k, (const SweepRec*)rec, gen_expl, n_gen_expl, u, k_b, k_1, du);       \
  } while (0)
      if (bs == 256) KIN_GEN_GO(256);
      else if (bs == 512) KIN_GEN_GO(512);
      else KIN_GEN_GO(1024);
#undef KIN_GEN_GO
      KIN_HIP(hipGetLastError());
      return;
    }
    if (adj) launch_sweep_t<true, true>(grid, smem, (int)N, (int)R, (int)P, (int)B, tile, 1, rec, pair_k, u, k_b, k_1, du, s);
    else launch_sweep_t<true, false>(grid, smem, (int)N, (int)R, (int)P, (int)B, tile, 1, rec, pair_k, u, k_b, k_1, du, s);
  } else {
    const int tile = 16 * 1024;   // 128 kB of accumulators per pass
    const int n_tiles = (int)ceil_div(N, tile);
    if (adj) launch_sweep_t<false, true>(grid, (size_t)tile * 8, (int)N, (int)R, (int)P, (int)B, tile, n_tiles, rec, pair_k, u, k_b, k_1, du, s);
    else launch_sweep_t<false, false>(grid, (size_t)tile * 8, (int)N, (int)R, (int)P, (int)B, tile, n_tiles, rec, pair_k, u, k_b, k_1, du, s);
  }
  KIN_HIP(hipGetLastError());
}

}  // namespace kin
