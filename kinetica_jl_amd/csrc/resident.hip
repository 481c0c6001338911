// Resident integrator (gfx950): one 512-thread workgroup owns one trajectory for the whole solve. The controller is
// resident_core.hpp, compiled here with a backend whose operations are workgroup-wide phases over the trajectory's
// vectors in global memory (L2-resident for the network sizes this path takes: up to ~2 000 species), separated by
// workgroup barriers where the multi-workgroup integrator of solver.cpp has kernel boundaries.
//   * Wavefront 0 runs the controller (its 64 lanes redundantly, on identical scalars); the other seven wait at a barrier
//     for its next command - the phase to run and its scalar arguments, posted in LDS - run that phase together with
//     wavefront 0 and come back. (First version: all 1 024 threads ran the controller redundantly. Its live scalars are
//     spilled per LANE around every phase call, and sixteen wavefronts doing that cost 70 us per step at 300 species.)
//   * The phases are NOT inlined into the controller (each is a function of its own with the whole register file; the
//     controller's live scalars are saved once per call instead of being spilled inside the phases' loops), and they take
//     their pointers from a context block in LDS and give them an address space: a pointer read from memory is generic to
//     the compiler, and generic accesses (flat_load / flat_store) cost this kernel a factor of ten in its first version.
//   * The LU cache's slot table lives in registers, slot i in lane i of every wavefront (reads = lane broadcasts, writes =
//     one predicated move): no shared copy, no hazard between wavefronts that run ahead of each other.
//   * The vectors every corrector iteration touches - y, d, psi, the error weights, the reaction rates and the window of
//     the Newton-matrix workspace that holds right-hand side, intermediate and solution of a solve (lu.cpp lays them out
//     contiguously) - live in LDS for the whole solve: a gather from them costs an LDS access instead of an L2 round trip,
//     and a stage's stores need no write acknowledgement before the barrier behind it. Factor values, index lists and the
//     difference history stay in global memory (L2).
//   * A whole corrector attempt (predictor, up to four iterations, their decisions) is ONE phase: between its iterations
//     nothing goes back to the controller.
//   * The dense Schur block is inverted by a blocked Gauss-Jordan (16 columns per block step, pivot block by the
//     single-wavefront inversion the multi-workgroup path uses, rank-16 updates on the matrix cores), ping-pong between
//     the slot's block and a scratch block of the trajectory.
// Reference: the work this replaces is init / solve! / reinit! of the reference's stiff solver and the chunk loop around
// it (src/solving/methods.jl:185-303, 717-865; solve_utils.jl:376-424, 435-509).
#include "resident.hpp"

// A phase is a function of its own (the whole register file, the controller's live scalars saved once per call) in the build
// with 256 registers per lane; in the 128-register build (resident_w4.hip) the calls' register saves go to scratch memory and
// inlining the phases is worth 13 % (300 species, 1 024 members: 4 200-4 600 -> 4 800-5 200 solves/s; no gain at 256 registers)
#ifndef RES_PHASE
#define RES_PHASE __device__ __noinline__
#endif

#include <atomic>

#include "exp_tab.hpp"
#include "gj_dev.hpp"
#include "segsum_dev.hpp"

#include <algorithm>
#include <new>

namespace kin {

namespace {

constexpr int RES_WG = 512, RES_WAVES = RES_WG / 64;

#define KIN_AS1 __attribute__((address_space(1)))
typedef KIN_AS1 double gd_t;
typedef KIN_AS1 const double gcd_t;
typedef KIN_AS1 const int32_t gci_t;
typedef KIN_AS1 const float gcf_t;
// Values read from the context block in LDS are the same in every lane, but the compiler keeps them in vector registers (a
// pointer = 2 VGPRs in each of 64 lanes; a gather plan = 34): readfirstlane moves them to scalar registers, where the global
// loads take them as a scalar base address - without it the gather loops of the solve spilled their pointers to scratch.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni(long long v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}
template <class T> __device__ __forceinline__ KIN_AS1 T* glob(T* p) { return (KIN_AS1 T*)(unsigned long long)uni((long long)(unsigned long long)p); }

// a gather plan with global-memory pointers
struct SegPlanViewG {
  gci_t *grp_off, *grp_dst, *grp_aux, *ell_a, *ell_b; gcf_t* ell_c;
  gci_t *seg_beg, *seg_end, *seg_dst, *seg_aux, *blk_beg, *blk_end, *blk_dst, *blk_aux, *long_a, *long_b; gcf_t* long_c;
  int32_t G, S, B, val_base, ell_total;
  int32_t n64, n32;   // medium rows (sorted longest first): [0, n64) have more than 64 entries, [n64, n32) 33 .. 64, the rest 9 .. 32
};
__device__ __forceinline__ SegPlanViewG plan_g(const SegPlanView& p, int n64 = -1, int n32 = -1) {
  SegPlanViewG g;
  g.grp_off = glob(p.grp_off); g.grp_dst = glob(p.grp_dst); g.grp_aux = glob(p.grp_aux);
  g.ell_a = glob(p.ell_a); g.ell_b = glob(p.ell_b); g.ell_c = glob(p.ell_c);
  g.seg_beg = glob(p.seg_beg); g.seg_end = glob(p.seg_end); g.seg_dst = glob(p.seg_dst); g.seg_aux = glob(p.seg_aux);
  g.blk_beg = glob(p.blk_beg); g.blk_end = glob(p.blk_end); g.blk_dst = glob(p.blk_dst); g.blk_aux = glob(p.blk_aux);
  g.long_a = glob(p.long_a); g.long_b = glob(p.long_b); g.long_c = glob(p.long_c);
  g.G = uni(p.G); g.S = uni(p.S); g.B = uni(p.B); g.val_base = uni(p.val_base); g.ell_total = uni(p.ell_total);
  g.n64 = n64 < 0 ? g.S : uni(n64); g.n32 = n32 < 0 ? g.S : uni(n32);   // unknown split: every medium row by a whole wavefront
  return g;
}
struct SegExtraG { const double* psi = nullptr; const double* d = nullptr; double cscal = 0.0; };   // (LDS vectors)
// ... and one whose TASK DESCRIPTORS (row ranges, destinations, aux positions: a few hundred integers per plan) sit in LDS: the
// first of a task's two dependent round trips to L2 becomes an LDS read (the three plans of every corrector iteration)
struct SegPlanViewL {
  const int32_t *grp_off, *grp_dst, *grp_aux; gci_t *ell_a, *ell_b; gcf_t* ell_c;
  const int32_t *seg_beg, *seg_end, *seg_dst, *seg_aux, *blk_beg, *blk_end, *blk_dst, *blk_aux; gci_t *long_a, *long_b; gcf_t* long_c;
  int32_t G, S, B, val_base, ell_total;
  int32_t n64, n32;
};

// hot gather plans kept in LDS (the per-round plans of the factorisation stay in global memory)
enum : int { PL_RHS = 0, PL_JAC, PL_RESID, PL_LZ, PL_NVU, PL_STAGEA, PL_STAGEC, PL_FWDZ, PL_FWD_DENSE, PL_BWDT, PL_BWDV, PL_COUNT };

// everything the phases need, written once by the kernel's prologue (LDS)
struct ResCtx {
  ResTrajDev T;
  const ResNetDev* net;
  SegPlanView plan[PL_COUNT];
  int32_t split[PL_COUNT][2];   // n64, n32 of the hot plans (counted once by the kernel's prologue)
  int32_t profile;
  int32_t l_y, l_d, l_psi, l_scale, l_win, l_rate;   // offsets (doubles) of the LDS-resident vectors in g_dyn
  int32_t desc_on, desc_off[3];                      // task descriptors of PL_RESID / PL_STAGEA / PL_STAGEC in g_dyn (offsets in int32)
  int64_t off_vec_end;
  int32_t N, R, nnzJ, ns, m, m16, mpad, nrounds, n_mono_ent, solve_mode, has_kmax, n_slots, rate_mode;
  int64_t off_diag, off_U, off_L, off_S, off_y, off_x, off_dinv, w_size;
  double k_max, t_mult;
};

struct ResShared {
  double red[RES_WAVES][8];
  double ru[36];
  double coef[8], gamma[8];
  double drift[RES_MAX_SLOTS], slot_c[RES_MAX_SLOTS];
  int slot_valid[RES_MAX_SLOTS];
  double pinv[2][16][17];   // pivot-block inverses of the dense Gauss-Jordan: current / next (look-ahead)
  long long prof[20];
  int bad;
  // command of the leader wavefront to the seven others (see resident_bdf_kernel)
  int cmd_op, cmd_i[3];
  long long cmd_l;
  double cmd_d[8];
  ResCorrIn corr;   // the corrector attempt being run (OP_CORRECTOR)
};

// LU-cache slot table (Solver's per-slot bookkeeping): only the leader wavefront touches it - its 64 lanes execute in
// lockstep, so a shared copy has no hazards, and lane i can look at slot i when a search runs over all slots
struct ResSlots {
  double c_fact[RES_MAX_SLOTS], crate[RES_MAX_SLOTS];
  long long crate_step[RES_MAX_SLOTS], crate_restart[RES_MAX_SLOTS], last_use[RES_MAX_SLOTS], jac_stamp[RES_MAX_SLOTS], step_stamp[RES_MAX_SLOTS];
  int valid[RES_MAX_SLOTS];
};

__shared__ ResCtx g_cx;
__shared__ ResShared g_sh;
__shared__ ResSlots g_sl;
__shared__ ResParams g_par;
// dynamic LDS: y | d | psi | scale | solve-vector window of W | rates (the dense inverse's row panel, 16 x (m16 + 1), reuses
// the rates' place: no rates are live during a factorisation)
extern __shared__ double g_dyn[];
__device__ __forceinline__ double* L_y() { return g_dyn + uni(g_cx.l_y); }
__device__ __forceinline__ double* L_d() { return g_dyn + uni(g_cx.l_d); }
__device__ __forceinline__ double* L_psi() { return g_dyn + uni(g_cx.l_psi); }
__device__ __forceinline__ double* L_scale() { return g_dyn + uni(g_cx.l_scale); }
__device__ __forceinline__ double* L_rate() { return g_dyn + uni(g_cx.l_rate); }
// the window, addressed with positions of the workspace W: L_win()[p] for off_y <= p < off_vec_end
__device__ __forceinline__ double* L_win() { return g_dyn + uni(g_cx.l_win) - uni((long long)g_cx.off_y); }

// phase kinds of the in-kernel profile (ResResult::prof, 10 ns ticks, thread 0's clock)
enum ProfId : int { PF_TOTAL = 0, PF_FACTOR = 1, PF_GJ = 2, PF_NEWTON = 3, PF_SOLVE = 4, PF_PREDICT = 5, PF_CHANGE_D = 6, PF_ACCEPT = 7,
                    PF_JAC = 8, PF_RHS = 9, PF_RESID = 10, PF_UPDATE = 11, PF_RATES = 12, PF_STAGEA = 13, PF_GEMV = 14, PF_STAGEC = 15,
                    PF_REDUCE = 16, PF_CTL_CORR = 17 };
struct ProfScope {   // (off unless KIN_RESIDENT_PROFILE is set: the clock reads cost ~0.3 us each)
  int id; long long t0;
  __device__ __forceinline__ ProfScope(int i) : id(i), t0(g_cx.profile ? wall_clock64() : 0) {}
  __device__ __forceinline__ ~ProfScope() { if (g_cx.profile && threadIdx.x == 0) g_sh.prof[id] += wall_clock64() - t0; }
};
#define RES_PROF(id) ProfScope prof_scope_##id(id)

__device__ __forceinline__ long long shfl_ll(long long v, int src) {
  const int lo = __shfl((int)(v & 0xffffffffll), src, 64), hi = __shfl((int)(v >> 32), src, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// Sums over lanes by DPP row shifts and row broadcasts (tools/wg_latency_probe.hip: five sums 0.26 us, by ds_bpermute
// 0.72 us - the LDS crossbar is shared by the CU's sixteen wavefronts). dpp_sum<64>: total in lane 63; <16> / <8>: the total of
// every group of 16 / 8 lanes in the group's last lane. Fixed order: bitwise reproducible.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int LANES>
__device__ __forceinline__ double dpp_sum(double v) {
  v += dpp_mov<0x111>(v);                    // row_shr:1
  v += dpp_mov<0x112>(v);                    // row_shr:2
  if (LANES >= 8) v += dpp_mov<0x114>(v);    // row_shr:4
  if (LANES >= 16) v += dpp_mov<0x118>(v);   // row_shr:8
  if (LANES >= 64) { v += dpp_mov<0x142>(v); v += dpp_mov<0x143>(v); }   // row_bcast:15, row_bcast:31
  return v;
}
// NV sums over the workgroup, the same value in every thread (fixed order: bitwise reproducible)
template <int NV>
__device__ __forceinline__ void wg_reduce(double (&v)[NV]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool pf = NV == 5 && g_cx.profile && threadIdx.x == 0;
  long long c0 = pf ? wall_clock64() : 0;
#pragma unroll
  for (int q = 0; q < NV; q++) v[q] = dpp_sum<64>(v[q]);
  if (lane == 63) {
#pragma unroll
    for (int q = 0; q < NV; q++) g_sh.red[wave][q] = v[q];
  }
  long long c1 = pf ? wall_clock64() : 0;
  __syncthreads();
  long long c2 = pf ? wall_clock64() : 0;
#pragma unroll
  for (int q = 0; q < NV; q++) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < RES_WAVES; w++) t += g_sh.red[w][q];
    v[q] = t;
  }
  __syncthreads();
  if (pf) { g_sh.prof[17] += c1 - c0; g_sh.prof[18] += c2 - c1; }      // (the rest of PF_REDUCE: reads + second barrier)
}

// The in-workgroup executor of a gather plan (kernels.hip: segsum_kernel spreads the same tasks over a grid): wavefront
// tasks round robin over the 8 wavefronts - long rows first (one wavefront walks such a row 512 entries per pass: no
// workgroup-wide reduction inside a phase), then medium rows, then the ELL groups of short rows.
// one group of LANES lanes per medium row (4 entries per lane), 64 / LANES rows per wavefront task
// srcA / srcB: arrays of the first / second factor (segsum_dev.hpp: seg_gather2), auxp: array of the `aux` operand
template <int OP, int LANES, class V, class SA, class SB, class OP_, class AP>
__device__ __forceinline__ void seg_rows_grouped(const V& p, SA srcA, SB srcB, OP_ out, AP auxp, const SegExtraG& ex, bool impl, int row0, int row_end) {
  const int lane = threadIdx.x & 63;
  const int g = lane / LANES, l = lane % LANES;
  const int sidx = row0 + g;
  const bool have = sidx < row_end;
  const int32_t e0 = have ? p.seg_beg[sidx] : 0, e1 = have ? p.seg_end[sidx] : 0;
  const int32_t dst = have ? p.seg_dst[sidx] : -1, aux = have ? p.seg_aux[sidx] : 0;
  const SegPre pre = seg_pre<OP>(out, auxp, l == LANES - 1 ? dst : -1, aux, ex);
  double acc = seg_gather2<OP, 4, false>(p, srcA, srcB, ex, impl, [&](int x) { const int32_t e = e0 + l + LANES * x; return e < e1 ? e : -1; });
  acc = dpp_sum<LANES>(acc);
  if (l == LANES - 1 && dst >= 0) seg_store<OP>(out, dst, acc, pre, ex);
}

template <int OP, class V, class SA, class SB, class OP_, class AP>
__device__ __forceinline__ void seg_run(const V& p, SA srcA, SB srcB, OP_ out, AP auxp, const SegExtraG& ex) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool impl = p.val_base >= 0;
  // task list: long rows, medium rows > 64 entries (a wavefront each), 33 .. 64 (four per wavefront), 9 .. 32 (eight per
  // wavefront), ELL groups of 64 short rows - most rows of the solve's stages have 9 .. 32 entries
  const int t1 = p.B, t2 = t1 + p.n64, t3 = t2 + ((p.n32 - p.n64 + 3) >> 2), t4 = t3 + ((p.S - p.n32 + 7) >> 3), ntask = t4 + p.G;
  for (int task = wave; task < ntask; task += RES_WAVES) {
    if (task < t1) {
      const int r = task;
      const int32_t e0 = p.blk_beg[r], e1 = p.blk_end[r], dst = p.blk_dst[r], aux = p.blk_aux[r];
      const SegPre pre = seg_pre<OP>(out, auxp, lane == 63 ? dst : -1, aux, ex);
      double acc = 0.0;
      for (int32_t base = e0; base < e1; base += 512)
        acc += seg_gather2<OP, 8, false>(p, srcA, srcB, ex, impl, [&](int x) { const int32_t e = base + lane + 64 * x; return e < e1 ? e : -1; });
      acc = dpp_sum<64>(acc);
      if (lane == 63) seg_store<OP>(out, dst, acc, pre, ex);
    } else if (task < t2) {
      seg_rows_grouped<OP, 64>(p, srcA, srcB, out, auxp, ex, impl, task - t1, p.n64);
    } else if (task < t3) {
      seg_rows_grouped<OP, 16>(p, srcA, srcB, out, auxp, ex, impl, p.n64 + 4 * (task - t2), p.n32);
    } else if (task < t4) {
      seg_rows_grouped<OP, 8>(p, srcA, srcB, out, auxp, ex, impl, p.n32 + 8 * (task - t3), p.S);
    } else {
      const int g = task - t4;
      const int32_t dst = p.grp_dst[g * 64 + lane], aux = p.grp_aux[g * 64 + lane];
      const int32_t c0 = p.grp_off[g], c1 = p.grp_off[g + 1];
      const SegPre pre = seg_pre<OP>(out, auxp, dst, aux, ex);
      double acc = 0.0;
      for (int32_t col = c0; col < c1; col += 8)
        acc += seg_gather2<OP, 8, true>(p, srcA, srcB, ex, impl, [&](int x) { return col + x < c1 ? (col + x) * 64 + lane : -1; });
      if (dst >= 0) seg_store<OP>(out, dst, acc, pre, ex);
    }
  }
}
// everything in one array (the factorisation's updates, the right-hand side, the Jacobian)
template <int OP, class V, class SP, class OP_>
__device__ __forceinline__ void seg_run1(const V& p, SP src, OP_ out, const SegExtraG& ex) { seg_run<OP>(p, src, src, out, src, ex); }

__device__ __forceinline__ SegPlanViewG hot_plan(int id) { return plan_g(g_cx.plan[id], g_cx.split[id][0], g_cx.split[id][1]); }
// the LDS-descriptor view of hot plan `id` (slot 0 / 1 / 2 = PL_RESID / PL_STAGEA / PL_STAGEC); layout of a slot, in int32:
// grp_off [G + 1] | grp_dst [64 G] | grp_aux [64 G] | seg_beg, seg_end, seg_dst, seg_aux [S each] | blk_beg, blk_end, blk_dst, blk_aux [B each]
__device__ __forceinline__ int desc_ints(const SegPlanView& p) { return (p.G + 1) + 128 * p.G + 4 * p.S + 4 * p.B; }
__device__ __forceinline__ SegPlanViewL lds_plan(int slot, int id) {
  const SegPlanViewG g = hot_plan(id);
  SegPlanViewL l;
  const int32_t* b = reinterpret_cast<const int32_t*>(g_dyn) + uni(g_cx.desc_off[slot]);
  l.grp_off = b; b += g.G + 1; l.grp_dst = b; b += 64 * g.G; l.grp_aux = b; b += 64 * g.G;
  l.seg_beg = b; b += g.S; l.seg_end = b; b += g.S; l.seg_dst = b; b += g.S; l.seg_aux = b; b += g.S;
  l.blk_beg = b; b += g.B; l.blk_end = b; b += g.B; l.blk_dst = b; b += g.B; l.blk_aux = b;
  l.ell_a = g.ell_a; l.ell_b = g.ell_b; l.ell_c = g.ell_c; l.long_a = g.long_a; l.long_b = g.long_b; l.long_c = g.long_c;
  l.G = g.G; l.S = g.S; l.B = g.B; l.val_base = g.val_base; l.ell_total = g.ell_total; l.n64 = g.n64; l.n32 = g.n32;
  return l;
}
// all threads: copies the descriptors of hot plan `id` into its LDS slot
__device__ __forceinline__ void stage_descriptors(int slot, int id) {
  const SegPlanViewG g = hot_plan(id);
  int32_t* b = reinterpret_cast<int32_t*>(g_dyn) + uni(g_cx.desc_off[slot]);
  auto put = [&](gci_t* src, int n) { for (int i = threadIdx.x; i < n; i += RES_WG) b[i] = src[i]; b += n; };
  put(g.grp_off, g.G + 1); put(g.grp_dst, 64 * g.G); put(g.grp_aux, 64 * g.G);
  put(g.seg_beg, g.S); put(g.seg_end, g.S); put(g.seg_dst, g.S); put(g.seg_aux, g.S);
  put(g.blk_beg, g.B); put(g.blk_end, g.B); put(g.blk_dst, g.B); put(g.blk_aux, g.B);
}

// ------------------------------------------------------------------------------------------------------------------
// phases (each called by all 512 threads; every one ends behind a barrier). y, d, psi, scale, the rates and the solve
// window are LDS arrays (L_y() ...), everything else global
// ------------------------------------------------------------------------------------------------------------------
enum VecOp : int { VO_LOAD_U0 = 0, VO_CHUNK_START_FROM_Y, VO_Y_FROM_CHUNK_START_CLIPPED, VO_Y_FROM_D0, VO_YTMP_FROM_D0, VO_YTMP_AXPY };
RES_PHASE void ph_vec(int op, double h0) {
  const int N = uni(g_cx.N), tid = threadIdx.x;
  double* y = L_y();
  if (op == VO_LOAD_U0) { gcd_t* u0 = glob(g_cx.T.u0); for (int i = tid; i < N; i += RES_WG) y[i] = u0[i]; }
  else if (op == VO_CHUNK_START_FROM_Y) { gd_t* cs = glob(g_cx.T.chunk_start); for (int i = tid; i < N; i += RES_WG) cs[i] = y[i]; }
  else if (op == VO_Y_FROM_CHUNK_START_CLIPPED) {
    gcd_t* cs = glob((const double*)g_cx.T.chunk_start);
    for (int i = tid; i < N; i += RES_WG) { const double v = cs[i]; y[i] = v < 0.0 ? 0.0 : v; }
  }
  else if (op == VO_Y_FROM_D0) { gcd_t* D = glob((const double*)g_cx.T.D); for (int i = tid; i < N; i += RES_WG) y[i] = D[i]; }
  else if (op == VO_YTMP_FROM_D0) { gcd_t* D = glob((const double*)g_cx.T.D); gd_t* yt = glob(g_cx.T.ytmp); for (int i = tid; i < N; i += RES_WG) yt[i] = D[i]; }
  else { gcd_t* f0 = glob((const double*)g_cx.T.f0); gd_t* yt = glob(g_cx.T.ytmp); for (int i = tid; i < N; i += RES_WG) yt[i] = y[i] + h0 * f0[i]; }
  __syncthreads();
}

RES_PHASE void ph_save_y(long long row, double time) {
  const int N = uni(g_cx.N);
  const double* y = L_y();
  gd_t* o = glob(g_cx.T.sol) + (size_t)row * N;
  for (int i = threadIdx.x; i < N; i += RES_WG) o[i] = y[i];
  if (threadIdx.x == 0) glob(g_cx.T.sol_t)[row] = time;
  __syncthreads();
}

RES_PHASE void ph_apply_rates(long long stop) {
  const int R = uni(g_cx.R);
  gd_t* k = glob(g_cx.T.k);
  const ResNetDev* net = g_cx.net;
  if (g_cx.rate_mode == 1) {
    gcd_t* src = glob(net->k_table) + (size_t)stop * R;
    for (int r = threadIdx.x; r < R; r += RES_WG) k[r] = src[r];
  } else if (g_cx.rate_mode == 2) {
    const double RT = 8.314462618 * glob(net->T_stops)[stop];
    gcd_t* Ea = glob(net->Ea); gcd_t* A = glob(net->A);
    const int has_kmax = g_cx.has_kmax; const double k_max = g_cx.k_max, t_mult = g_cx.t_mult;
    for (int r = threadIdx.x; r < R; r += RES_WG) k[r] = arrhenius_one(Ea[r], A[r], RT, has_kmax, k_max, t_mult);
  }
  __syncthreads();
}

// mass-action rates of state u into the LDS rate array (make_rs, solve_utils.jl:318-334); no barrier inside
template <class UP>
__device__ __forceinline__ void rates_into(UP u) {
  const int R = uni(g_cx.R);
  gci_t* x0 = glob(g_cx.net->x0); gci_t* x1 = glob(g_cx.net->x1);
  gcd_t* k = glob((const double*)g_cx.T.k);
  double* rate = L_rate();
  for (int r = threadIdx.x; r < R; r += RES_WG) {
    const int32_t a = x0[r], b = x1[r];
    const double ub = b >= 0 ? u[b] : 1.0;
    rate[r] = k[r] * u[a] * ub;
  }
}

enum RhsOp : int { RO_Y_TO_F0 = 0, RO_YTMP_TO_F1, RO_YTMP_TO_F0 };
RES_PHASE void ph_rhs(int op) {
  RES_PROF(PF_RHS);
  gd_t* out = glob(op == RO_YTMP_TO_F1 ? g_cx.T.f1 : g_cx.T.f0);
  if (op == RO_Y_TO_F0) rates_into((const double*)L_y());
  else rates_into(glob((const double*)g_cx.T.ytmp));
  __syncthreads();
  const double* rate = L_rate();
  seg_run<SEG_COEF_SET>(hot_plan(PL_RHS), rate, rate, out, rate, SegExtraG{});
  __syncthreads();
}

// analytic Jacobian at y into T.jv (CSR values)
RES_PHASE void ph_jac() {
  RES_PROF(PF_JAC);
  const int R = uni(g_cx.R);
  const double* u = L_y();
  gci_t* x0 = glob(g_cx.net->x0); gci_t* x1 = glob(g_cx.net->x1);
  gcd_t* k = glob((const double*)g_cx.T.k); gd_t* dr = glob(g_cx.T.dr);
  for (int r = threadIdx.x; r < R; r += RES_WG) {
    const int32_t a = x0[r], b = x1[r];
    const double kk = k[r];
    double d0, d1 = 0.0;
    if (b < 0) d0 = kk;
    else if (b == a) d0 = 2.0 * kk * u[a];
    else { d0 = kk * u[b]; d1 = kk * u[a]; }
    dr[2 * r] = d0; dr[2 * r + 1] = d1;
  }
  __syncthreads();
  seg_run1<SEG_COEF_SET>(hot_plan(PL_JAC), glob((const double*)g_cx.T.dr), glob(g_cx.T.jv), SegExtraG{});
  __syncthreads();
}

RES_PHASE ResNorms ph_norms(bool with_f1, double atol, double rtol) {
  const int N = uni(g_cx.N);
  const double* y = L_y();
  gcd_t* f0p = glob((const double*)g_cx.T.f0); gcd_t* f1p = glob((const double*)g_cx.T.f1);
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  double vm = 0.0;   // max |f0| / (0.1 |y| + w): the reciprocal of CVODE's upper bound on the first step (cvUpperBoundH0)
  for (int i = threadIdx.x; i < N; i += RES_WG) {
    const double y0 = y[i], f0 = f0p[i];
    const double sc = atol + rtol * fabs(y0);
    const double a = y0 / sc, b = f0 / sc;
    v[0] += a * a; v[1] += b * b;
    vm = fmax(vm, fabs(f0) / (0.1 * fabs(y0) + sc));
    if (!isfinite(f0)) v[3] = 1.0;
    if (with_f1) { const double f1 = f1p[i]; const double c = (f1 - f0) / sc; v[2] += c * c; if (!isfinite(f1)) v[3] = 1.0; }
  }
  wg_reduce<4>(v);
  // (a restart's reduction, not a step's: plain shuffles)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) vm = fmax(vm, __shfl_xor(vm, off, 64));
  if ((threadIdx.x & 63) == 0) g_sh.red[threadIdx.x >> 6][0] = vm;
  __syncthreads();
  double dmax = 0.0;
#pragma unroll
  for (int w = 0; w < RES_WAVES; w++) dmax = fmax(dmax, g_sh.red[w][0]);
  __syncthreads();
  const double Nd = (double)N;
  return ResNorms{sqrt(v[0] / Nd), sqrt(v[1] / Nd), sqrt(v[2] / Nd), dmax, v[3] > 0.0 ? 1 : 0};
}

RES_PHASE void ph_init_D(bool from_ytmp, double h) {
  const int N = uni(g_cx.N);
  gcd_t* yt = glob((const double*)g_cx.T.ytmp);
  const double* y = L_y();
  gcd_t* f0 = glob((const double*)g_cx.T.f0);
  gd_t* D = glob(g_cx.T.D);
  for (int i = threadIdx.x; i < N; i += RES_WG) {
    D[i] = from_ytmp ? yt[i] : y[i];
    D[(size_t)N + i] = f0[i] * h;
#pragma unroll
    for (int j = 2; j < RES_D_ROWS; j++) D[(size_t)j * N + i] = 0.0;
  }
  __syncthreads();
}

// predictor from the backward differences (gamma in g_sh.gamma): y, psi, d = 0, scale; barrier behind it
__device__ __forceinline__ void predict_body(int order, double alpha_o, double atol, double rtol) {
  const int N = uni(g_cx.N);
  gcd_t* D = glob((const double*)g_cx.T.D);
  double* y = L_y(); double* psi = L_psi(); double* d = L_d(); double* scale = L_scale();
  for (int i = threadIdx.x; i < N; i += RES_WG) {
    double yp = D[i], ps = 0.0;
    for (int j = 1; j <= order; j++) {
      const double dj = D[(size_t)j * N + i];
      yp += dj;
      ps += dj * g_sh.gamma[j];
    }
    y[i] = yp;
    psi[i] = ps / alpha_o;
    d[i] = 0.0;
    scale[i] = atol + rtol * fabs(yp);
  }
  __syncthreads();
}
RES_PHASE void ph_predict(int order, double alpha_o, double atol, double rtol) {
  RES_PROF(PF_PREDICT);
  predict_body(order, alpha_o, atol, rtol);
}

// D[0..ord] <- (R U)^T D[0..ord], matrix in g_sh.ru
RES_PHASE void ph_change_D(int ord) {
  RES_PROF(PF_CHANGE_D);
  const int N = uni(g_cx.N);
  gd_t* D = glob(g_cx.T.D);
  for (int i = threadIdx.x; i < N; i += RES_WG) {
    double v[6], o[6];
#pragma unroll
    for (int j = 0; j < 6; j++) v[j] = j <= ord ? D[(size_t)j * N + i] : 0.0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < 6; q++) t += (q <= ord ? g_sh.ru[q * 6 + a] : 0.0) * v[q];
      o[a] = t;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) if (j <= ord) D[(size_t)j * N + i] = o[j];
  }
  __syncthreads();
}

RES_PHASE void ph_accept(int order) {
  RES_PROF(PF_ACCEPT);
  const int N = uni(g_cx.N);
  gd_t* D = glob(g_cx.T.D);
  const double* d = L_d();
  for (int i = threadIdx.x; i < N; i += RES_WG) {
    const double di = d[i];
    D[(size_t)(order + 2) * N + i] = di - D[(size_t)(order + 1) * N + i];
    D[(size_t)(order + 1) * N + i] = di;
    double carry = di;
    for (int j = order; j >= 0; j--) {
      carry += D[(size_t)j * N + i];
      D[(size_t)j * N + i] = carry;
    }
  }
  __syncthreads();
}

// dense output into solution row `row` (weights in g_sh.coef)
RES_PHASE void ph_interp(int order, long long row) {
  const int N = uni(g_cx.N);
  gcd_t* D = glob((const double*)g_cx.T.D);
  gd_t* o = glob(g_cx.T.sol) + (size_t)row * N;
  for (int i = threadIdx.x; i < N; i += RES_WG) {
    double v = D[i];
    for (int j = 1; j <= order; j++) v += g_sh.coef[j] * D[(size_t)j * N + i];
    o[i] = v;
  }
  __syncthreads();
}

// drift guard of the LU cache at a restart (solver_kernels.hip: slot_drift_kernel): g_sh.drift[s] for every valid slot
// (validity and c_fact of the slots in g_sh.slot_valid / slot_c, written by the caller)
RES_PHASE void ph_drift() {
  const int N = uni(g_cx.N), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  gcd_t* jv = glob((const double*)g_cx.T.jv);
  gci_t* j_diag = glob(g_cx.net->j_diag);
  for (int s = wave; s < g_cx.n_slots; s += RES_WAVES) {
    if (!g_sh.slot_valid[s]) { if (lane == 0) g_sh.drift[s] = 0.0; continue; }
    const double c = g_sh.slot_c[s];
    gcd_t* jd = glob((const double*)g_cx.T.jd) + (size_t)s * N;
    double worst = 1.0;
    for (int i = lane; i < N; i += 64) {
      const double m_old = 1.0 - c * jd[i], m_new = 1.0 - c * jv[j_diag[i]];
      const double q = m_old / m_new;
      const double dev = (q > 0.0) ? fmax(q, 1.0 / q) : 1e300;
      worst = fmax(worst, dev == dev ? dev : 1e300);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_down(worst, off, 64));
    if (lane == 0) g_sh.drift[s] = worst - 1.0;
  }
  __syncthreads();
}

// ---- dense inverse: blocked Gauss-Jordan, 16 columns per block step, X -> Y ping-pong (no tile reads what another wavefront
// of the same step writes). Block step kb with K = rows / columns 16 kb .. 16 kb + 15:
//     P = X[K,K]^-1                     (wavefront 0: gj_inv16_wave, result in g_sh.pinv)
//     RP = P X[K,:]                     (row panel into LDS g_dyn, 16 x m16; RP[:,K] = P)
//     Y[K,:] = RP ;  Y[i,j] = X[i,j] - X[i,K] RP[:,j]  (j not in K) ;  Y[i,K] = -X[i,K] P        (i not in K)
// v_mfma_f64_16x16x4_f64 operand layout (tools/mfma_probe.hip): A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k,
// D[i][j] in lane 16 (i % 4) + j, register i / 4.
typedef double res_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gj_blocked(gd_t* S, gd_t* S2, int ld, int m16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int nb = m16 / 16, ldr = m16 + 1;
  double* rp = L_rate();   // (no rates are live during a factorisation)
  gd_t* X = S;
  gd_t* Y = S2;
  // Look-ahead: the LAST wavefront does not take row strips; during the update of step kb it forms the next pivot block
  // Y[K+1, K+1] itself and inverts it - the 16 dependent pivots of gj_inv16_wave (2.7 us on one wavefront) then run next to the
  // other wavefronts' tile updates instead of in front of them. Used when giving that wavefront up does not lengthen the update
  // (the strips spread over 7 wavefronts in as many rounds as over 8). The pivot block it inverts is, bit for bit, the tile its
  // owner writes to Y.
  const int nw = ((nb + RES_WAVES - 2) / (RES_WAVES - 1) == (nb + RES_WAVES - 1) / RES_WAVES) ? RES_WAVES - 1 : RES_WAVES;
  const bool look = nw < RES_WAVES;
  auto invert_into = [&](double (*P)[17], double (&a)[4]) {       // one wavefront; a = the block in (row lane >> 2, 4 columns) layout
    const int r = lane >> 2, c0 = (lane & 3) * 4;
    const bool vanished = gj_inv16_wave(a, lane);
#pragma unroll
    for (int x = 0; x < 4; x++) P[r][c0 + x] = a[x];
    if (vanished) g_sh.bad = 1;
  };
  if (look) {
    if (wave == 0) {
      const int r = lane >> 2, c0 = (lane & 3) * 4;
      double a[4];
#pragma unroll
      for (int x = 0; x < 4; x++) a[x] = X[(size_t)r * ld + c0 + x];
      invert_into(g_sh.pinv[0], a);
    }
    __syncthreads();
  }
  for (int kb = 0; kb < nb; kb++) {
    const int K0 = kb * 16;
    double (*P)[17] = g_sh.pinv[look ? (kb & 1) : 0];
    if (!look) {
      if (wave == 0) {
        const int r = lane >> 2, c0 = (lane & 3) * 4;
        double a[4];
#pragma unroll
        for (int x = 0; x < 4; x++) a[x] = X[(size_t)(K0 + r) * ld + K0 + c0 + x];
        invert_into(P, a);
      }
      __syncthreads();
    }
    // row panel: wavefront w forms the 16 x 16 tiles J = w, w + 8, ...
    for (int J = wave; J < nb; J += RES_WAVES) {
      res_d4 acc = {0.0, 0.0, 0.0, 0.0};
      if (J != kb) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
          const double ap = P[li][4 * ks + lk];
          const double bx = X[(size_t)(K0 + 4 * ks + lk) * ld + J * 16 + li];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, bx, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int i = 4 * v + lk;
        rp[i * ldr + J * 16 + li] = (J == kb) ? P[i][li] : acc[v];
      }
    }
    __syncthreads();
    if (look && wave == RES_WAVES - 1) {
      // the next pivot block after this step, formed like any tile of strip kb + 1, then inverted
      const int I = kb + 1;
      if (I < nb) {
        double (*Pn)[17] = g_sh.pinv[I & 1];
        double a[4], xo[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) a[ks] = X[(size_t)(I * 16 + li) * ld + K0 + 4 * ks + lk];
#pragma unroll
        for (int v = 0; v < 4; v++) xo[v] = X[(size_t)(I * 16 + 4 * v + lk) * ld + I * 16 + li];
        res_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], rp[(4 * ks + lk) * ldr + I * 16 + li], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; v++) Pn[4 * v + lk][li] = xo[v] - acc[v];
        __builtin_amdgcn_wave_barrier();
        const int r = lane >> 2, c0 = (lane & 3) * 4;
        double b4[4];
#pragma unroll
        for (int x = 0; x < 4; x++) b4[x] = Pn[r][c0 + x];
        __builtin_amdgcn_wave_barrier();
        invert_into(Pn, b4);
      }
    } else {
      // update: wavefront w takes the row strips I = w, w + nw, ...; its column-panel tile X[I,K] is loaded once per strip
      for (int I = wave; I < nb; I += nw) {
        if (I == kb) {
          for (int J = 0; J < nb; J++)
#pragma unroll
            for (int v = 0; v < 4; v++) Y[(size_t)(K0 + 4 * v + lk) * ld + J * 16 + li] = rp[(4 * v + lk) * ldr + J * 16 + li];
          continue;
        }
        double a[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) a[ks] = X[(size_t)(I * 16 + li) * ld + K0 + 4 * ks + lk];
        for (int J = 0; J < nb; J++) {
          double xo[4];
#pragma unroll
          for (int v = 0; v < 4; v++) xo[v] = (J == kb) ? 0.0 : X[(size_t)(I * 16 + 4 * v + lk) * ld + J * 16 + li];
          res_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], rp[(4 * ks + lk) * ldr + J * 16 + li], acc, 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 4; v++) Y[(size_t)(I * 16 + 4 * v + lk) * ld + J * 16 + li] = xo[v] - acc[v];
        }
      }
    }
    __syncthreads();
    gd_t* t = X; X = Y; Y = t;
  }
  if (X != S) {   // odd number of block steps: the inverse sits in the scratch block
    for (int idx = threadIdx.x; idx < m16 * m16; idx += RES_WG) { const int i = idx / m16, j = idx - i * m16; S[(size_t)i * ld + j] = X[(size_t)i * ld + j]; }
    __syncthreads();
  }
}

// M = I - c J factorised into slot `slot` (SparseLU::factor); returns true when a pivot vanished
RES_PHASE bool ph_factor(int slot, double c, bool keep_diag) {
  RES_PROF(PF_FACTOR);
  const int tid = threadIdx.x;
  const ResNetDev* net = g_cx.net;
  gd_t* W = glob(g_cx.T.W) + (size_t)slot * (size_t)g_cx.w_size;
  gcd_t* Wc = (gcd_t*)W;
  gcd_t* jv = glob((const double*)g_cx.T.jv);
  const int m = g_cx.m, mpad = g_cx.mpad, nnzJ = g_cx.nnzJ, ns = g_cx.ns;
  const long long off_S = g_cx.off_S, off_L = g_cx.off_L, off_diag = g_cx.off_diag, off_dinv = g_cx.off_dinv;
  if (tid == 0) g_sh.bad = 0;
  for (long long e = tid; e < g_cx.off_y; e += RES_WG) W[e] = 0.0;
  __syncthreads();
  {
    gci_t* jmap = glob(net->jmap);
    const int total = nnzJ + (mpad - m);
    for (int e = tid; e < total; e += RES_WG) {
      if (e < nnzJ) {
        const int32_t jm = jmap[e];
        W[jm & 0x7fffffff] = (jm < 0 ? 1.0 : 0.0) - c * jv[e];
      } else {
        const int d = m + (e - nnzJ);
        W[off_S + (long long)d * mpad + d] = 1.0;
      }
    }
  }
  __syncthreads();
  {
    gci_t* ent_pivot = glob(net->ent_pivot);
    for (int r = 0; r < g_cx.nrounds; r++) {
      const int e0 = net->round_e0[r], e1 = net->round_e0[r + 1];
      for (int e = e0 + tid; e < e1; e += RES_WG) {
        const double w = W[off_L + e], piv = W[off_diag + ent_pivot[e]];
        const double l = w / piv;
        if (!(fabs(piv) >= PIVOT_MIN) || (w != 0.0 && !(fabs(l) <= PIVOT_GROWTH_MAX))) g_sh.bad = 1;
        W[off_L + e] = l;
      }
      __syncthreads();
      seg_run1<SEG_PROD_SUB>(plan_g(net->schur[r]), Wc, W, SegExtraG{});
      __syncthreads();
    }
  }
  if (g_cx.solve_mode != RES_SOLVE_PLAIN) {
    for (int i = tid; i < ns; i += RES_WG) W[off_dinv + i] = 1.0 / W[off_diag + i];
    __syncthreads();
    gci_t* mep = glob(net->mono_ent_ptr); gci_t* mp = glob(net->mono_ptr); gci_t* mf = glob(net->mono_fac); gci_t* md = glob(net->mono_dst);
    gcf_t* ms = glob(net->mono_sign);
    for (int e = tid; e < g_cx.n_mono_ent; e += RES_WG) {
      double acc = 0.0;
      for (int32_t mo = mep[e]; mo < mep[e + 1]; mo++) {
        double prod = (double)ms[mo];
        for (int32_t f = mp[mo]; f < mp[mo + 1]; f++) prod *= W[mf[f]];
        acc += prod;
      }
      W[md[e]] = acc;
    }
    __syncthreads();
    if (g_cx.solve_mode == RES_SOLVE_FUSED) {
      seg_run1<SEG_PROD_AUXSUB>(hot_plan(PL_LZ), Wc, W, SegExtraG{});
      seg_run1<SEG_PROD_NEG>(hot_plan(PL_NVU), Wc, W, SegExtraG{});
      __syncthreads();
    }
  }
  if (m > 0) { RES_PROF(PF_GJ); gj_blocked(W + off_S, glob(g_cx.T.gj_scratch), mpad, g_cx.m16); }
  if (keep_diag) {
    gd_t* jd = glob(g_cx.T.jd) + (size_t)slot * g_cx.N;
    gci_t* j_diag = glob(net->j_diag);
    for (int i = tid; i < g_cx.N; i += RES_WG) jd[i] = jv[j_diag[i]];
  }
  __syncthreads();
  const bool bad = g_sh.bad != 0;
  __syncthreads();
  return bad;
}

// x = S^-1 y2; y2 and x in the LDS window. A group of LANES lanes takes a row with all of the row's loads in flight; LANES is
// chosen so that the workgroup covers the rows in as few passes as possible (m <= 128: 4 lanes per row, one pass - with 16 lanes
// per row m = 108 took four passes of seven loads per lane: 2.7 us per solve against 1.x now; one wavefront per row, as
// solver_kernels.hip's gemv_kernel has it, is seven dependent rounds)
template <int LANES>
__device__ __forceinline__ void gemv_rows(gcd_t* S, int ld, int m, const double* y2, double* x) {
  const int l = threadIdx.x % LANES, grp = threadIdx.x / LANES;
  for (int row0 = 0; row0 < m; row0 += RES_WG / LANES) {
    const int row = row0 + grp;
    double acc = 0.0;
    if (row < m) {
      gcd_t* a = S + (size_t)row * ld;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int j = l;
      for (; j + 3 * LANES < m; j += 4 * LANES) {
        const double p0 = a[j], p1 = a[j + LANES], p2 = a[j + 2 * LANES], p3 = a[j + 3 * LANES];
        a0 += p0 * y2[j]; a1 += p1 * y2[j + LANES]; a2 += p2 * y2[j + 2 * LANES]; a3 += p3 * y2[j + 3 * LANES];
      }
      double q0 = 0.0, q1 = 0.0, q2 = 0.0;       // up to three more columns, their loads issued together
      if (j < m) q0 = a[j] * y2[j];
      if (j + LANES < m) q1 = a[j + LANES] * y2[j + LANES];
      if (j + 2 * LANES < m) q2 = a[j + 2 * LANES] * y2[j + 2 * LANES];
      acc = ((a0 + a1) + (a2 + a3)) + ((q0 + q1) + q2);
    }
    acc = dpp_sum<LANES>(acc);
    if (l == LANES - 1 && row < m) x[row] = acc;
  }
  __syncthreads();
}
__device__ __forceinline__ void gemv_wg(gcd_t* S, int ld, int m, const double* y2, double* x) {
  if (m <= RES_WG / 4) gemv_rows<4>(S, ld, m, y2, x);
  else if (m <= RES_WG / 8 * 2) gemv_rows<8>(S, ld, m, y2, x);
  else gemv_rows<16>(S, ld, m, y2, x);
}

// M x = b with the factors in W (global) and the vectors in the LDS window: b at win[yloc], x at win[xloc] (SparseLU::solve).
// One function per solve form: an iteration only ever fetches the code of the form its network uses.
RES_PHASE void solve_fused(gcd_t* Wc) {
  const SegExtraG ex{};
  double* win = L_win();
  const double* winc = win;
  const int m = uni(g_cx.m), mpad = uni(g_cx.mpad), ns = uni(g_cx.ns);
  const long long off_y = uni((long long)g_cx.off_y), off_x = uni((long long)g_cx.off_x);
  const bool ld = uni(g_cx.desc_on) != 0;
  { RES_PROF(PF_STAGEA);
    if (ld) seg_run<SEG_PROD_AUXSUB>(lds_plan(1, PL_STAGEA), Wc, winc, win, winc, ex); else seg_run<SEG_PROD_AUXSUB>(hot_plan(PL_STAGEA), Wc, winc, win, winc, ex);
    __syncthreads(); }
  { RES_PROF(PF_GEMV); gemv_wg(Wc + uni((long long)g_cx.off_S), mpad, m, winc + off_y + ns, win + off_x); }
  { RES_PROF(PF_STAGEC);
    if (ld) seg_run<SEG_PROD_SET>(lds_plan(2, PL_STAGEC), Wc, winc, win, winc, ex); else seg_run<SEG_PROD_SET>(hot_plan(PL_STAGEC), Wc, winc, win, winc, ex);
    __syncthreads(); }
}
RES_PHASE void solve_explicit(gcd_t* Wc) {
  const SegExtraG ex{};
  double* win = L_win();
  const double* winc = win;
  const int m = uni(g_cx.m), mpad = uni(g_cx.mpad), ns = uni(g_cx.ns);
  const long long off_y = uni((long long)g_cx.off_y), off_x = uni((long long)g_cx.off_x);
  seg_run<SEG_PROD_AUXSUB>(hot_plan(PL_FWDZ), Wc, winc, win, winc, ex); __syncthreads();
  if (m > 0) {
    seg_run<SEG_PROD_SUB>(hot_plan(PL_FWD_DENSE), Wc, winc, win, winc, ex); __syncthreads();
    gemv_wg(Wc + uni((long long)g_cx.off_S), mpad, m, winc + off_y + ns, win + off_x);
  }
  seg_run<SEG_PROD_AUXSUB>(hot_plan(PL_BWDT), Wc, winc, win, winc, ex); __syncthreads();
  seg_run<SEG_PROD_SET>(hot_plan(PL_BWDV), Wc, winc, win, winc, ex); __syncthreads();
}
RES_PHASE void solve_plain(gcd_t* Wc) {
  // plain substitution: the divisor of a backward row (aux) is a factor value, i.e. in global memory
  const SegExtraG ex{};
  const ResNetDev* net = g_cx.net;
  double* win = L_win();
  const double* winc = win;
  const int m = uni(g_cx.m), mpad = uni(g_cx.mpad), ns = uni(g_cx.ns), nrounds = uni(g_cx.nrounds);
  const long long off_y = uni((long long)g_cx.off_y), off_x = uni((long long)g_cx.off_x);
  for (int r = 1; r < nrounds; r++) { seg_run<SEG_PROD_SUB>(plan_g(net->fwd[r]), Wc, winc, win, winc, ex); __syncthreads(); }
  if (m > 0) {
    if (ns > 0) { seg_run<SEG_PROD_SUB>(hot_plan(PL_FWD_DENSE), Wc, winc, win, winc, ex); __syncthreads(); }
    gemv_wg(Wc + uni((long long)g_cx.off_S), mpad, m, winc + off_y + ns, win + off_x);
  }
  for (int r = nrounds - 1; r >= 0; r--) { seg_run<SEG_PROD_SUB_DIV>(plan_g(net->bwd[r]), Wc, winc, win, Wc, ex); __syncthreads(); }
}
__device__ __forceinline__ void solve_wg(gcd_t* Wc) {
  const int mode = uni(g_cx.solve_mode);
  if (mode == RES_SOLVE_FUSED) solve_fused(Wc);
  else if (mode == RES_SOLVE_EXPLICIT) solve_explicit(Wc);
  else solve_plain(Wc);
}

// one corrector iteration: residual, solve, update, the five sums of the decision and the error test
// (solver_kernels.hip: rates_skip_kernel, segsum<SEG_COEF_BDF>, the solve, bdf_newton_kernel)
__device__ __forceinline__ ResSums newton_body(int slot, double c, double upd, int order, double ec, double ec_m, double ec_p, double atol, double rtol) {
  gcd_t* Wc = glob((const double*)g_cx.T.W) + (size_t)slot * (size_t)g_cx.w_size;
  double* y = L_y(); double* d = L_d();
  double* win = L_win();
  {
    RES_PROF(PF_RESID);
    { RES_PROF(PF_RATES); rates_into((const double*)y); __syncthreads(); }
    SegExtraG ex;
    ex.psi = L_psi(); ex.d = d; ex.cscal = c;
    const double* rate = L_rate();
    if (uni(g_cx.desc_on) != 0) seg_run<SEG_COEF_BDF>(lds_plan(0, PL_RESID), rate, rate, win, rate, ex);
    else seg_run<SEG_COEF_BDF>(hot_plan(PL_RESID), rate, rate, win, rate, ex);
    __syncthreads();
  }
  { RES_PROF(PF_SOLVE); solve_wg(Wc); }
  RES_PROF(PF_UPDATE);
  const int N = uni(g_cx.N);
  gci_t* xloc = glob(g_cx.net->xloc);
  const double* scale = L_scale();
  gcd_t* D = glob((const double*)g_cx.T.D);
  double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < N; i += RES_WG) {
    const double dy = upd * win[xloc[i]];
    const double q = dy / scale[i];
    v[0] += q * q;
    const double yy = y[i] + dy, dd = d[i] + dy;
    const double sce = atol + rtol * fabs(yy);
    if (yy < 0.0) v[4] = fmax(v[4], yy < -RES_NEG_DEEP * sce ? RES_NEG_MARK : 1.0);
    const double e = ec * dd / sce;
    v[1] += e * e + (isfinite(yy) ? 0.0 : INFINITY);
    if (order > 1) { const double em = ec_m * (D[(size_t)order * N + i] + dd) / sce; v[2] += em * em; }
    if (order < RES_MAX_ORDER) { const double ep = ec_p * (dd - D[(size_t)(order + 1) * N + i]) / sce; v[3] += ep * ep; }
    y[i] = yy; d[i] = dd;
  }
  { RES_PROF(PF_REDUCE); wg_reduce<5>(v); }
  return ResSums{v[0], v[1], v[2], v[3], v[4]};
}

// a whole corrector attempt: predictor + iterations + their decisions (resident_core.hpp: res_corrector_loop), by all wavefronts
struct InnerOps {
  __device__ void predict_inner(int order, const double*, double alpha_o, double atol, double rtol) {
    RES_PROF(PF_PREDICT);
    predict_body(order, alpha_o, atol, rtol);
  }
  __device__ ResSums newton_iter_inner(int slot, double c, double upd, int order, double ec, double ec_m, double ec_p, double atol, double rtol) {
    return newton_body(slot, c, upd, order, ec, ec_m, ec_p, atol, rtol);
  }
};
RES_PHASE ResAttempt ph_corrector() {
  RES_PROF(PF_NEWTON);
  const ResCorrIn in = g_sh.corr;
  InnerOps ops;
  return res_corrector_loop(ops, in, g_sh.gamma, g_cx.N);
}

// ------------------------------------------------------------------------------------------------------------------
// the backend handed to the controller (wavefront 0): posts a phase to the other wavefronts and runs it with them; the slot
// table of the LU cache in its registers
// ------------------------------------------------------------------------------------------------------------------
enum CmdOp : int { OP_EXIT = 0, OP_VEC, OP_SAVE_Y, OP_APPLY_RATES, OP_RHS, OP_JAC, OP_NORMS, OP_INIT_D, OP_PREDICT, OP_CHANGE_D, OP_ACCEPT,
                   OP_INTERP, OP_DRIFT, OP_FACTOR, OP_CORRECTOR };
struct DevBackend {
  __device__ int lane() const { return threadIdx.x & 63; }
  __device__ int n_species() const { return g_cx.N; }
  __device__ void profile_out(int64_t* out) const {
    for (int i = 0; i < 20; i++) out[i] = g_sh.prof[i];   // written by thread 0 only: this wavefront's own stores
  }

  // ---- slot table (g_sl)
  __device__ double slot_c_fact(int i) const { return g_sl.c_fact[i]; }
  __device__ double slot_crate(int i) const { return g_sl.crate[i]; }
  __device__ long long slot_crate_step(int i) const { return g_sl.crate_step[i]; }
  __device__ long long slot_crate_restart(int i) const { return g_sl.crate_restart[i]; }
  __device__ void slot_touch(int i, long long clock) { g_sl.last_use[i] = clock; }
  __device__ void slot_rate(int i, double crate, long long step, long long restart) { g_sl.crate[i] = crate; g_sl.crate_step[i] = step; g_sl.crate_restart[i] = restart; }
  __device__ void slot_drop(int i) { g_sl.valid[i] = 0; }
  __device__ void slot_made(int i, double c, long long clock, long long jac_stamp, long long step_stamp) {
    g_sl.c_fact[i] = c; g_sl.crate[i] = 1.0; g_sl.valid[i] = 1; g_sl.last_use[i] = clock; g_sl.jac_stamp[i] = jac_stamp; g_sl.step_stamp[i] = step_stamp;
  }
  __device__ void slots_invalidate(bool reset) {
    g_sl.valid[lane()] = 0;
    if (reset) { g_sl.c_fact[lane()] = 0.0; g_sl.last_use[lane()] = 0; g_sl.crate[lane()] = 1.0; g_sl.crate_step[lane()] = 0; g_sl.crate_restart[lane()] = -1;
                 g_sl.jac_stamp[lane()] = 0; g_sl.step_stamp[lane()] = 0; }
  }
  // slot whose c_fact is closest (in ratio) to c and within the band, lowest index on ties; -1: none (Solver::nearest_slot)
  __device__ int nearest_slot(double c, double band, long long n_restarts, long long max_age) const {
    const double cf = g_sl.c_fact[lane()];
    const bool ok = lane() < g_cx.n_slots && g_sl.valid[lane()] && n_restarts - g_sl.jac_stamp[lane()] <= max_age && fabs(c / cf - 1.0) <= band;
    double r = ok ? fabs(log(c / cf)) : 1e300;
    if (!(r < 1e300)) r = 1e300;
    int idx = lane();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double r2 = __shfl_xor(r, off, 64);
      const int i2 = __shfl_xor(idx, off, 64);
      if (r2 < r || (r2 == r && i2 < idx)) { r = r2; idx = i2; }
    }
    return r < 1e300 ? idx : -1;
  }
  // a slot for a new factorisation: the first unused or expired one, else the least recently used (Solver::victim_slot)
  __device__ int victim_slot(long long n_restarts, long long max_age, int n_slots) const {
    const bool in = lane() < n_slots;
    const bool free_ = in && (!g_sl.valid[lane()] || n_restarts - g_sl.jac_stamp[lane()] > max_age);
    const unsigned long long fm = __ballot(free_);
    if (fm) return __ffsll((long long)fm) - 1;
    long long u = in ? g_sl.last_use[lane()] : 0x7fffffffffffffffll;
    int idx = lane();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const long long u2 = shfl_ll(u, lane() ^ off);
      const int i2 = __shfl_xor(idx, off, 64);
      if (u2 < u || (u2 == u && i2 < idx)) { u = u2; idx = i2; }
    }
    return idx;
  }
  // ---- commands: lane 0 posts the phase and its scalar arguments, the barrier releases the other wavefronts into it
  __device__ void post(int op, int i0 = 0, int i1 = 0, int i2 = 0, long long l0 = 0, double d0 = 0.0, double d1 = 0.0, double d2 = 0.0,
                       double d3 = 0.0, double d4 = 0.0, double d5 = 0.0, double d6 = 0.0) {
    if (threadIdx.x == 0) {
      g_sh.cmd_op = op; g_sh.cmd_i[0] = i0; g_sh.cmd_i[1] = i1; g_sh.cmd_i[2] = i2; g_sh.cmd_l = l0;
      g_sh.cmd_d[0] = d0; g_sh.cmd_d[1] = d1; g_sh.cmd_d[2] = d2; g_sh.cmd_d[3] = d3; g_sh.cmd_d[4] = d4; g_sh.cmd_d[5] = d5; g_sh.cmd_d[6] = d6;
    }
    __syncthreads();
  }
  // drift guard at a restart: slots whose diag(I - c_s J) moved by more than `max_drift` are dropped; returns how many
  __device__ int drift_check(double max_drift) {
    g_sh.slot_valid[lane()] = g_sl.valid[lane()]; g_sh.slot_c[lane()] = g_sl.c_fact[lane()];
    post(OP_DRIFT);
    ph_drift();
    const bool drop = lane() < g_cx.n_slots && g_sl.valid[lane()] && !(g_sh.drift[lane()] <= max_drift);
    if (drop) g_sl.valid[lane()] = 0;
    return __popcll(__ballot(drop));
  }

  // ---- vectors
  __device__ void vec(int op, double h0) { post(OP_VEC, op, 0, 0, 0, h0); ph_vec(op, h0); }
  __device__ void load_u0() { vec(VO_LOAD_U0, 0.0); }
  __device__ void chunk_start_from_y() { vec(VO_CHUNK_START_FROM_Y, 0.0); }
  __device__ void y_from_chunk_start_clipped() { vec(VO_Y_FROM_CHUNK_START_CLIPPED, 0.0); }
  __device__ void y_from_D0() { vec(VO_Y_FROM_D0, 0.0); }
  __device__ void ytmp_from_D0() { vec(VO_YTMP_FROM_D0, 0.0); }
  __device__ void ytmp_axpy(double h0) { vec(VO_YTMP_AXPY, h0); }
  __device__ void save_y(long long row, double time) { post(OP_SAVE_Y, 0, 0, 0, row, time); ph_save_y(row, time); }
  __device__ void set_time(long long row, double time) { if (threadIdx.x == 0) glob(g_cx.T.sol_t)[row] = time; }
  __device__ void apply_rates(long long stop) { post(OP_APPLY_RATES, 0, 0, 0, stop); ph_apply_rates(stop); }
  __device__ void rhs_y_to_f0() { post(OP_RHS, RO_Y_TO_F0); ph_rhs(RO_Y_TO_F0); }
  __device__ void rhs_ytmp_to_f1() { post(OP_RHS, RO_YTMP_TO_F1); ph_rhs(RO_YTMP_TO_F1); }
  __device__ void rhs_ytmp_to_f0() { post(OP_RHS, RO_YTMP_TO_F0); ph_rhs(RO_YTMP_TO_F0); }
  __device__ void eval_jac_y() { post(OP_JAC); ph_jac(); }
  __device__ ResNorms norms(bool with_f1, double atol, double rtol) { post(OP_NORMS, with_f1 ? 1 : 0, 0, 0, 0, atol, rtol); return ph_norms(with_f1, atol, rtol); }
  __device__ void init_D(bool from_ytmp, double h) { post(OP_INIT_D, from_ytmp ? 1 : 0, 0, 0, 0, h); ph_init_D(from_ytmp, h); }
  __device__ void predict(int order, const double* gamma, double alpha_o, double atol, double rtol) {
    (void)gamma;   // the kernel's prologue put the coefficients into g_sh.gamma
    post(OP_PREDICT, order, 0, 0, 0, alpha_o, atol, rtol);
    ph_predict(order, alpha_o, atol, rtol);
  }
  __device__ void change_D(int ord, const double (*RU)[6]) {
    if (threadIdx.x < 36) g_sh.ru[threadIdx.x] = RU[threadIdx.x / 6][threadIdx.x % 6];
    post(OP_CHANGE_D, ord);
    ph_change_D(ord);
  }
  __device__ void accept(int order) { post(OP_ACCEPT, order); ph_accept(order); }
  __device__ void interp(int order, const double* p, long long row) {
    if (threadIdx.x <= RES_MAX_ORDER) g_sh.coef[threadIdx.x] = (int)threadIdx.x <= order ? p[threadIdx.x] : 0.0;
    post(OP_INTERP, order, 0, 0, row);
    ph_interp(order, row);
  }
  __device__ bool factor(int slot, double c, bool keep_diag) { post(OP_FACTOR, slot, keep_diag ? 1 : 0, 0, 0, c); return ph_factor(slot, c, keep_diag); }
  __device__ ResAttempt corrector(const ResCorrIn& in, const double*) {
    if (threadIdx.x == 0) g_sh.corr = in;
    post(OP_CORRECTOR);
    return ph_corrector();
  }
};

// what the seven other wavefronts do: wait for a command, run its phase, until the leader posts OP_EXIT
__device__ void worker_loop() {
  // (profile: what the first worker wavefront spends waiting for the next command = the controller's own time between
  // two phases, incl. the calls' register saves and the posting)
  const bool pf = g_cx.profile && threadIdx.x == 64;
  for (;;) {
    const long long w0 = pf ? wall_clock64() : 0;
    __syncthreads();
    if (pf) g_sh.prof[19] += wall_clock64() - w0;
    const int op = g_sh.cmd_op;
    if (op == OP_EXIT) return;
    const int i0 = g_sh.cmd_i[0], i1 = g_sh.cmd_i[1];
    const long long l0 = g_sh.cmd_l;
    const double d0 = g_sh.cmd_d[0], d1 = g_sh.cmd_d[1], d2 = g_sh.cmd_d[2];
    switch (op) {
      case OP_VEC: ph_vec(i0, d0); break;
      case OP_SAVE_Y: ph_save_y(l0, d0); break;
      case OP_APPLY_RATES: ph_apply_rates(l0); break;
      case OP_RHS: ph_rhs(i0); break;
      case OP_JAC: ph_jac(); break;
      case OP_NORMS: (void)ph_norms(i0 != 0, d0, d1); break;
      case OP_INIT_D: ph_init_D(i0 != 0, d0); break;
      case OP_PREDICT: ph_predict(i0, d0, d1, d2); break;
      case OP_CHANGE_D: ph_change_D(i0); break;
      case OP_ACCEPT: ph_accept(i0); break;
      case OP_INTERP: ph_interp(i0, l0); break;
      case OP_DRIFT: ph_drift(); break;
      case OP_FACTOR: (void)ph_factor(i0, d0, i1 != 0); break;
      case OP_CORRECTOR: (void)ph_corrector(); break;
      default: break;
    }
  }
}

// counts of medium rows with more than 64 / more than 32 entries (the rows are sorted longest first); all threads
__device__ __forceinline__ void count_splits(int id) {
  const SegPlanView& p = g_cx.plan[id];
  gci_t* sb = glob(p.seg_beg); gci_t* se = glob(p.seg_end);
  int c64 = 0, c32 = 0;
  for (int r = threadIdx.x; r < p.S; r += RES_WG) { const int len = se[r] - sb[r]; c64 += len > 64; c32 += len > 32; }
  for (int off = 32; off >= 1; off >>= 1) { c64 += __shfl_down(c64, off, 64); c32 += __shfl_down(c32, off, 64); }
  if ((threadIdx.x & 63) == 0 && (c64 | c32)) { atomicAdd(&g_cx.split[id][0], c64); atomicAdd(&g_cx.split[id][1], c32); }
}

__shared__ __attribute__((aligned(16))) unsigned char g_ctl_mem[sizeof(ResidentBdf<DevBackend>)];

// Two builds of this file (Makefile: resident.hip as it is, resident_w4.hip = this file with RES_WAVES_PER_EU 4): the register
// budget of the kernel AND of every phase it calls follows from the kernel's waves-per-SIMD attribute.
//   2: 256 VGPRs per lane, one workgroup per compute unit - the fastest single trajectory;
//   4: 128 VGPRs (a few spills in the corrector, the factorisation and the order change), TWO workgroups share a compute
//      unit and hide each other's latencies: an ensemble of more members than compute units gets 20-40 % more solves/s at
//      300 species (4 000-4 600 against 3 300-3 600), a single member is 10-15 % slower (resident.cpp picks per launch).
#ifndef RES_WAVES_PER_EU
#define RES_WAVES_PER_EU 2
#endif
__global__ __launch_bounds__(RES_WG) __attribute__((amdgpu_waves_per_eu(RES_WAVES_PER_EU, RES_WAVES_PER_EU))) void resident_bdf_kernel(const ResNetDev* __restrict__ net_p, const ResTrajDev* __restrict__ traj,
                                                               const ResParams* __restrict__ par_p) {
  if (threadIdx.x == 0) {
    const ResNetDev& n = *net_p;
    g_par = *par_p;
    g_cx.T = traj[blockIdx.x];
    g_cx.net = net_p;
    g_cx.plan[PL_RHS] = n.rhs_plan; g_cx.plan[PL_JAC] = n.jac_plan; g_cx.plan[PL_RESID] = n.resid_plan;
    g_cx.plan[PL_LZ] = n.lz_build; g_cx.plan[PL_NVU] = n.nvu_build; g_cx.plan[PL_STAGEA] = n.stageA; g_cx.plan[PL_STAGEC] = n.stageC;
    g_cx.plan[PL_FWDZ] = n.fwdZ; g_cx.plan[PL_FWD_DENSE] = n.fwd_dense; g_cx.plan[PL_BWDT] = n.bwdT; g_cx.plan[PL_BWDV] = n.bwdV;
    for (int i = 0; i < PL_COUNT; i++) { g_cx.split[i][0] = 0; g_cx.split[i][1] = 0; }
    g_cx.profile = par_p->profile;
    g_cx.N = n.N; g_cx.R = n.R; g_cx.nnzJ = n.nnzJ; g_cx.ns = n.ns; g_cx.m = n.m; g_cx.m16 = (n.m + 15) / 16 * 16; g_cx.mpad = n.mpad;
    g_cx.nrounds = n.nrounds; g_cx.n_mono_ent = n.n_mono_ent; g_cx.solve_mode = n.solve_mode; g_cx.has_kmax = n.has_kmax;
    g_cx.n_slots = par_p->n_slots; g_cx.rate_mode = par_p->rate_mode;
    g_cx.off_diag = n.off_diag; g_cx.off_U = n.off_U; g_cx.off_L = n.off_L; g_cx.off_S = n.off_S; g_cx.off_y = n.off_y; g_cx.off_x = n.off_x;
    g_cx.off_dinv = n.off_dinv; g_cx.w_size = n.w_size;
    g_cx.k_max = n.k_max; g_cx.t_mult = n.t_mult;
    g_cx.off_vec_end = n.off_vec_end;
    const int win = (int)(n.off_vec_end - n.off_y);
    g_cx.l_y = 0; g_cx.l_d = n.N; g_cx.l_psi = 2 * n.N; g_cx.l_scale = 3 * n.N; g_cx.l_win = 4 * n.N; g_cx.l_rate = 4 * n.N + win;
    // task descriptors of the corrector's three plans behind the vectors (the host provisioned the LDS for them or did not)
    g_cx.desc_on = n.desc_in_lds;
    const int m16_ = (n.m + 15) / 16 * 16;
    const int tail = n.R > 16 * (m16_ + 1) ? n.R : 16 * (m16_ + 1);
    int off = 2 * (4 * n.N + win + tail);       // in int32
    g_cx.desc_off[0] = off; off += desc_ints(n.resid_plan);
    g_cx.desc_off[1] = off; off += desc_ints(n.stageA);
    g_cx.desc_off[2] = off;
  }
  if (threadIdx.x < 20) g_sh.prof[threadIdx.x] = 0;
  __syncthreads();
  for (int id = 0; id < PL_COUNT; id++) count_splits(id);
  __syncthreads();
  if (g_cx.desc_on) { stage_descriptors(0, PL_RESID); stage_descriptors(1, PL_STAGEA); stage_descriptors(2, PL_STAGEC); __syncthreads(); }
  if (threadIdx.x >= 64) { worker_loop(); return; }
  // the leader wavefront: controller state and parameters live in LDS (one wavefront in lockstep: no hazards), not in
  // registers that would be spilled around every phase call
  const long long t_begin = wall_clock64();
  const long long c_begin = clock64();
  DevBackend b;
  ResidentBdf<DevBackend>* ctl = new (g_ctl_mem) ResidentBdf<DevBackend>(b, g_par);
  if (threadIdx.x <= RES_MAX_ORDER) g_sh.gamma[threadIdx.x] = ctl->gamma[threadIdx.x];   // (ordered by the first command's barrier)
  ResResult r = ctl->run();
  b.post(OP_EXIT);
  r.prof[PF_TOTAL] = wall_clock64() - t_begin;
  (void)c_begin;
  if (threadIdx.x == 0) *g_cx.T.result = r;
}

}  // namespace

#if RES_WAVES_PER_EU == 2
size_t resident_dyn_lds(int N, int R, int m, int64_t window) {
  const int m16 = (m + 15) / 16 * 16;
  return ((size_t)4 * N + (size_t)window + (size_t)std::max(R, 16 * (m16 + 1))) * sizeof(double);
}
size_t resident_desc_bytes(const SegPlanView& resid, const SegPlanView& stageA, const SegPlanView& stageC) {
  auto ints = [](const SegPlanView& p) { return (size_t)(p.G + 1) + 128 * (size_t)p.G + 4 * (size_t)p.S + 4 * (size_t)p.B; };
  return (ints(resid) + ints(stageA) + ints(stageC) + 2) / 2 * 2 * sizeof(int32_t);
}
size_t resident_static_lds() {
  hipFuncAttributes a{};
  KIN_HIP(hipFuncGetAttributes(&a, (const void*)resident_bdf_kernel));
  return a.sharedSizeBytes;
}
#define RES_LAUNCH_NAME launch_resident
#else
#define RES_LAUNCH_NAME launch_resident_shared_cu
#endif

void RES_LAUNCH_NAME(int K, size_t dyn_lds, const ResNetDev* d_net, const ResTrajDev* d_traj, const ResParams* d_par, hipStream_t s) {
  if (K <= 0) return;
  // the attribute belongs to the (function, device) pair: handles of one process may live on different GPUs
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  KIN_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    KIN_HIP(hipFuncSetAttribute((const void*)resident_bdf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS_BUDGET));
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(resident_bdf_kernel, dim3((unsigned)K), dim3(RES_WG), dyn_lds, s, d_net, d_traj, d_par);
  KIN_HIP(hipGetLastError());
}

}  // namespace kin
