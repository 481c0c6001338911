// Direct solver for the Newton matrix M = I - c*J of the implicit BDF step, replacing the
// KLU sparse LU the reference's documented solver uses (CVODE_BDF(linear_solver=:KLU),
// docs/src/getting-started.md:69).
//
// CRN Jacobians are "a sparse tail plus a few hub species". A classical fill-reducing LU has
// long dependency chains in its triangular solves, which is latency poison on a GPU, so the
// ordering here is chosen for depth instead:
//   * species with many neighbours (hubs) are never eliminated sparsely;
//   * the remaining tail is eliminated in a few ROUNDS, each round a maximal independent set
//     of low-degree nodes of the current elimination graph (1x1 pivots, all independent, so a
//     round is one parallel gather pass with no intra-round dependencies);
//   * whatever is left (hubs + the stubborn core) forms one dense Schur block that is inverted
//     explicitly (blocked Gauss-Jordan), so its share of every solve is a single GEMV.
// Factorisation = rounds x (scale L, gather Schur updates) + dense inverse (+ the explicit inverses
// of the sparse triangular blocks and the products of the fused solve, see below); a solve = three
// dependent kernels (gather | GEMV | gather), five without the fused products, 2*rounds+3 by plain
// substitution - instead of hundreds of levels. All updates are pull-style gathers over
// precomputed index lists (SegPlan), i.e. deterministic and atomic-free. Pivoting is static
// (diagonal), like KLU's refactor path, and guarded: a vanishing pivot raises a device flag that
// the integrator answers with a fresh Jacobian and a shorter step (M -> I as h -> 0).
#pragma once
#include <algorithm>
#include <map>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"
#include "solver_kernels.hpp"

namespace kin {

struct LUOptions {
  int hub_degree = 32;       // >= this many neighbours in J+J^T: straight to the dense block
  int max_rounds = 8;
  int max_tail_degree = 8;   // candidate filter: non-hub neighbours at elimination time
  int max_degree = 60;       // candidate filter: all neighbours at elimination time
  int min_round = 8;         // a round with fewer pivots than this ends the sparse elimination (it no longer pays for its two
                             // dependent launches; inside the resident integrator a round is two barriers, and 2 is enough)
};

struct SparseLU {
  // structure (new = elimination order, old = species index)
  int32_t n = 0, ns = 0, m = 0, mpad = 0, nrounds = 0;
  int64_t nnzU = 0;          // stored U entries of the sparse rows (L has the same count)
  int64_t schur_macs = 0;
  std::vector<int32_t> perm, iperm, round_ptr;
  // layout of the single value array W (so every gather indexes one base pointer)
  int64_t off_diag = 0, off_U = 0, off_L = 0, off_S = 0, off_y = 0, off_x = 0, off_vec_end = 0, w_size = 0;

  // Factor values live in SLOTS: one value array W = [diag | U | L | dense Schur block | solve vectors] plus the
  // ping-pong copy of the Schur block per slot. The symbolic structure (plans, maps) is shared. Several slots = the LU
  // cache of solver.cpp: factorisations for different c = h / alpha_k stay resident (HBM is plentiful: 25 MB per slot at
  // 10k species, 157 MB at 50k) and are reused across step-size changes and across restarts.
  struct Slot {
    DevBuf<double> W, S2;
    DevBuf<double> jd;                      // diag(J) of the Jacobian behind this factorisation (drift test of the LU cache)
    const double* sinv = nullptr;           // where the inverse of the Schur block ended up (inside W or S2)
    double c_fact = 0.0;
    double crate = 1.0;                     // contraction rate this factorisation has shown in the corrector (1 = unknown; CVODE's crate)
    int64_t crate_step = 0, crate_restart = -1;   // accepted-step / restart counters when that rate was last MEASURED
    int64_t last_use = 0;
    int64_t jac_stamp = 0;                  // restart counter at the time the Jacobian behind this factorisation was evaluated
    int64_t step_stamp = 0;                 // accepted-step counter at that time
    bool valid = false;
  };
  std::vector<Slot> slots;
  DevBuf<double> pinv;                      // Gauss-Jordan pivot block inverses (scratch of a factorisation)
  size_t slot_bytes() const { return ((size_t)w_size + (size_t)std::max(mpad, 64) * std::max(mpad, 64)) * sizeof(double); }
  void ensure_slots(int n, hipStream_t s);  // allocates (and zeroes) slots up to n
  DevBuf<int32_t> jmap;                     // J entry -> W position (bit 31: diagonal)
  DevBuf<int32_t> ent_pivot;                // sparse entry -> its pivot
  DevBuf<int32_t> yloc, xloc;               // species -> position of its rhs / solution in W
  std::vector<int32_t> ent_ptr;             // host: entries of pivot p are [ent_ptr[p], ent_ptr[p+1])
  std::vector<SegPlanDev> schur, fwd, bwd;  // per round
  SegPlanDev fwd_dense;
  // Explicit inverses of the sparse triangular blocks. The sparse-to-sparse parts L11 / U11 of the factors are tiny (7 k
  // entries at 10k species; the bulk of L and U couples the sparse pivots to the dense block) and shallow (depth = rounds),
  // so their inverses Z = L11^-1, V = U11^-1 are formed entry by entry at every factorisation (each entry = a short sum
  // of products along the elimination DAG, one small launch) and a solve becomes
  //   y1 = Z b1 | y2 = b2 - L21 y1 | x2 = S^-1 y2 | t = y1 - U12 x2 | x1 = V t
  // = 5 dependent launches instead of 2 * rounds + 3 (15 at 6 rounds): the Newton iteration is a latency chain.
  bool explicit_tri = false;
  int64_t off_Z = 0, off_V = 0, off_dinv = 0, off_y1 = 0, off_t = 0, nnzZ = 0, nnzV = 0, n_monomials = 0;
  int32_t n_mono_ent = 0;
  DevBuf<int32_t> mono_ent_ptr, mono_ptr, mono_fac, mono_dst;
  DevBuf<float> mono_sign;
  SegPlanDev fwdZ, bwdT, bwdV;
  // Fused solve: with LZ = L21 * L11^-1 and NVU = -(U11^-1 * U12) formed at every factorisation (two more gather
  // launches, entries = short sums of products of factor values) the solve is THREE dependent launches:
  //   [y1 = Z b1 ; y2 = b2 - LZ b1] | x2 = S^-1 y2 | x1 = V y1 + NVU x2
  bool fused_tri = false;
  int64_t off_LZ = 0, off_NVU = 0, off_zero = 0, off_VA = 0, off_VC = 0, nnzLZ = 0, nnzNVU = 0, n_fused_products = 0;
  SegPlanDev lz_build, nvu_build, stageA, stageC;
  int64_t nnzJ = 0;

  void analyze(int32_t n, const std::vector<int32_t>& j_ptr, const std::vector<int32_t>& j_col,
               const LUOptions& opt, hipStream_t s);
  // host_only: the symbolic analysis without a device (nothing is uploaded or allocated; sizes and plan statistics only) -
  // kin_lu_analyze_host, used to choose elimination parameters and by the CPU tests
  bool host_only = false;
  int64_t plan_entries = 0, plan_tasks = 0, plan_long_rows = 0;   // summed over every gather plan built (host_only statistics)
  // keep_host: host copies of everything analyze() would upload, keyed by the address of the device-side member (the CPU
  // replay of the factorisation in tests/native/ reads them; test infrastructure - the product never sets it)
  bool keep_host = false;
  std::map<const void*, std::vector<int32_t>> host_i32;
  std::map<const void*, std::vector<float>> host_f32;
  std::map<const void*, SegPlanHost> host_plan;
  void up(DevBuf<int32_t>& d, const std::vector<int32_t>& h, hipStream_t s) { if (keep_host) host_i32[&d] = h; if (!host_only) d.upload(h, s); }
  void up(DevBuf<float>& d, const std::vector<float>& h, hipStream_t s) { if (keep_host) host_f32[&d] = h; if (!host_only) d.upload(h, s); }
  void up(SegPlanDev& d, const SegPlanHost& h, hipStream_t s) {
    plan_entries += (int64_t)std::max(h.ell_a.size(), h.ell_b.size()) + (int64_t)std::max(h.long_a.size(), h.long_b.size());
    plan_tasks += h.n_groups() + h.n_segs(); plan_long_rows += h.n_blks();
    if (keep_host) host_plan[&d] = h;
    if (!host_only) d.upload(h, s);
  }
  void sync(hipStream_t s) { if (!host_only) KIN_HIP(hipStreamSynchronize(s)); }
  // M = I - c*J, factorised into slot `slot`
  // `bad`: device flag raised when a pivot vanishes (a multiplier exceeds 1e8 in magnitude or is not finite): pivoting
  // is static (diagonal), so the caller answers with a fresh Jacobian and a shorter step
  void factor(double c, const double* d_jvals, int slot, int* bad, hipStream_t s) { factor_into(c, d_jvals, slots[slot], pinv.p, bad, s); }
  // the same into a slot that lives outside this object (ensemble.cpp: every member of an ensemble has slots of its own, the
  // symbolic tables are shared); `pinv_scratch`: 2 x 32 x 32 doubles, one per stream that factorises concurrently
  void factor_into(double c, const double* d_jvals, Slot& q, double* pinv_scratch, int* bad, hipStream_t s);
  void factor_sparse_into(double c, const double* d_jvals, Slot& q, int* bad, hipStream_t s);
  void alloc_slot(Slot& q, hipStream_t s) const;   // value arrays of one slot, zeroed
  // solves M x = b in place with the factors of `slot`: b was written to W[yloc[v]], x is read from W[xloc[v]]
  // (W = that slot's array). `skip`: optional device flag making every kernel of the solve a no-op.
  void solve(const int* skip, int slot, hipStream_t s);
  // fused_tri only: the same solve with the corrector update of the BDF step folded into its last launch (the right-hand
  // side was written to W[yloc], `f` carries what the update and its decision need; f.skip makes every launch a no-op)
  void solve_newton(int slot, NewtonFuse f, hipStream_t s);
  int newton_grid() const { return stagec_newton_grid(stageC.view(), m); }   // workgroups of that last launch
  DevBuf<int32_t> x2_species;               // species behind the dense block's rows
};

// dense / LU helper kernels (solver_kernels.hip)
void launch_lu_assemble(int64_t nnzJ, const int32_t* jmap, const double* jvals, double c, double* W,
                        int64_t off_S, int32_t m, int32_t mpad, hipStream_t s);
void launch_lu_scale(int64_t e0, int64_t e1, const int32_t* ent_pivot, double* W, int64_t off_L, int64_t off_diag, int* bad, hipStream_t s);
void launch_lu_recip(int n, const double* diag, double* dinv, hipStream_t s);
void launch_lu_mono(int n_ent, const int32_t* ent_ptr, const int32_t* mono_ptr, const int32_t* fac, const float* sign,
                    const int32_t* dst, double* W, hipStream_t s);
double* launch_gauss_jordan(double* S, double* S2, int32_t mpad, double* pinv, int* bad, hipStream_t s);
// the same for up to GJ_BMAX matrices of one size in one chain of launches (blockIdx.z = matrix)
constexpr int GJ_BMAX = 16;
struct GjBatch { double* X[GJ_BMAX]; double* Y[GJ_BMAX]; double* pinv[GJ_BMAX]; double* pinv_next[GJ_BMAX]; int* bad[GJ_BMAX]; };
int launch_gauss_jordan_batched(int n, double* const* S, double* const* S2, int32_t mpad, double* pinv, int* const* bad, hipStream_t s);
void launch_gemv(const double* S, int32_t ld, int32_t m, const double* y, double* x, const int* skip, hipStream_t s);

}  // namespace kin
