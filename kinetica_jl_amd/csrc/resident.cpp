// Host-side owner of the resident integrator (resident.hip / resident_core.hpp): symbolic analysis, table upload,
// per-trajectory workspaces, ONE launch per solve (or per ensemble of solves) and the read-back of its result block.
// Takes the networks whose trajectory fits one workgroup (up to a few thousand species): at those sizes the host-driven
// integrator of solver.cpp spends its time in launch calls and dependency gaps (~75 us per step whatever N is), which is
// the regime of the reference's documented CRNs (docs/src/getting-started.md:43-70) and of BASELINE configs[0..1].
#include "resident.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "handle.hpp"
#include "lu.hpp"
#include "resident_setup.hpp"
#include "solver.hpp"

namespace kin {

struct ResidentSolver {
  kin_network* h;
  SparseLU lu;
  SegPlanDev resid_plan;
  DevBuf<int32_t> d_jdiag, d_round_dummy;
  ResNetDev hn{};
  DevBuf<ResNetDev> d_net;
  bool ok = false;
  std::string why;
  size_t dyn_lds = 0;
  // workspaces (grown on demand)
  int K_cap = 0, n_slots = 0;
  size_t per_traj = 0;
  DevBuf<double> work, Wbuf, jdbuf, gjbuf, d_u0, d_sol, d_solt, d_save, d_tstops, d_Tstops;
  DevBuf<ResTrajDev> d_traj;
  DevBuf<ResResult> d_res;
  DevBuf<ResParams> d_par;
  std::vector<ResTrajDev> h_traj;

  explicit ResidentSolver(kin_network* hh) : h(hh) {
    const NetworkHost& H = h->host;
    hipStream_t s = h->stream;
    LUOptions opt;
    opt.min_round = 2;   // a round costs two barriers here, not two dependent launches
    opt.max_rounds = std::min(opt.max_rounds, RES_MAX_ROUNDS);
    lu.analyze((int32_t)H.N, H.j_ptr, H.j_col, opt, s);
    if (lu.m > RES_MAX_DENSE) { why = "dense Schur block beyond the resident integrator's limit"; return; }
    if (lu.nrounds > RES_MAX_ROUNDS) { why = "too many elimination rounds"; return; }
    dyn_lds = resident_dyn_lds((int)H.N, (int)H.R, lu.m, lu.off_vec_end - lu.off_y);
    if (dyn_lds > RES_LDS_BUDGET) { why = "state, rates and solve vectors do not fit one compute unit's LDS"; return; }
    std::vector<int32_t> yl(H.N), ident(H.N);
    lu.yloc.download(yl.data(), H.N, s);
    KIN_HIP(hipStreamSynchronize(s));
    for (int64_t i = 0; i < H.N; i++) ident[i] = (int32_t)i;
    resid_plan.upload(build_seg_plan(H.N, H.sp_ptr.data(), yl.data(), H.sp_rxn.data(), nullptr, H.sp_coef.data(), false, ident.data()), s);
    d_jdiag.upload(H.j_diag, s);
    hn.N = (int32_t)H.N; hn.R = (int32_t)H.R; hn.nnzJ = (int32_t)H.nnz();
    hn.ns = lu.ns; hn.m = lu.m; hn.mpad = lu.mpad; hn.nrounds = lu.nrounds; hn.n_mono_ent = lu.n_mono_ent;
    hn.solve_mode = lu.fused_tri ? RES_SOLVE_FUSED : (lu.explicit_tri ? RES_SOLVE_EXPLICIT : RES_SOLVE_PLAIN);
    hn.off_diag = lu.off_diag; hn.off_U = lu.off_U; hn.off_L = lu.off_L; hn.off_S = lu.off_S; hn.off_y = lu.off_y; hn.off_x = lu.off_x;
    hn.off_dinv = lu.off_dinv; hn.off_vec_end = lu.off_vec_end; hn.w_size = lu.w_size;
    hn.x0 = h->x0.p; hn.x1 = h->x1.p; hn.jmap = lu.jmap.p; hn.ent_pivot = lu.ent_pivot.p; hn.yloc = lu.yloc.p; hn.xloc = lu.xloc.p;
    hn.j_diag = d_jdiag.p;
    hn.mono_ent_ptr = lu.mono_ent_ptr.p; hn.mono_ptr = lu.mono_ptr.p; hn.mono_fac = lu.mono_fac.p; hn.mono_dst = lu.mono_dst.p;
    hn.mono_sign = lu.mono_sign.p;
    for (int r = 0; r <= lu.nrounds; r++) hn.round_e0[r] = lu.ent_ptr[lu.round_ptr[r]];
    hn.rhs_plan = h->rhs_plan.view(); hn.jac_plan = h->jac_plan.view(); hn.resid_plan = resid_plan.view();
    hn.lz_build = lu.lz_build.view(); hn.nvu_build = lu.nvu_build.view(); hn.stageA = lu.stageA.view(); hn.stageC = lu.stageC.view();
    // LDS for the task descriptors of the corrector's three gather plans, where it is to be had: not beyond the kernel's budget,
    // and not at the price of the second workgroup per compute unit an ensemble would get without them
    {
      const size_t desc = resident_desc_bytes(hn.resid_plan, hn.stageA, hn.stageC);
      const size_t static_lds = resident_static_lds();
      const bool two_before = 2 * (dyn_lds + static_lds) <= (size_t)160 * 1024, two_after = 2 * (dyn_lds + desc + static_lds) <= (size_t)160 * 1024;
      hn.desc_in_lds = (lu.fused_tri && dyn_lds + desc <= RES_LDS_BUDGET && (two_after || !two_before)) ? 1 : 0;
      if (hn.desc_in_lds) dyn_lds += desc;
    }
    hn.fwdZ = lu.fwdZ.view(); hn.fwd_dense = lu.fwd_dense.view(); hn.bwdT = lu.bwdT.view(); hn.bwdV = lu.bwdV.view();
    for (int r = 0; r < lu.nrounds; r++) { hn.schur[r] = lu.schur[r].view(); hn.fwd[r] = lu.fwd[r].view(); hn.bwd[r] = lu.bwd[r].view(); }
    // the multi-workgroup factorisation's slot (SparseLU::analyze allocates one) is not used by this path
    lu.slots.clear();
    ok = true;
  }

  size_t slot_bytes() const { return ((size_t)lu.w_size + (size_t)h->host.N) * sizeof(double); }

  // (re)allocates the workspaces of K trajectories with `slots` LU-cache slots each
  void ensure(int K, int slots, int64_t sol_rows, bool own_sol) {
    const NetworkHost& H = h->host;
    hipStream_t s = h->stream;
    auto al = [](size_t x) { return (x + 7) / 8 * 8; };
    const size_t N = (size_t)H.N, R = (size_t)H.R;
    per_traj = al(R) + al((size_t)RES_D_ROWS * N) + 8 * al(N) + al((size_t)H.nnz()) + al(R) + al(2 * R + 2);
    if (K > K_cap || slots != n_slots) {
      work.alloc((size_t)K * per_traj);
      Wbuf.alloc((size_t)K * slots * (size_t)lu.w_size);
      jdbuf.alloc((size_t)K * slots * N);
      gjbuf.alloc((size_t)K * std::max<size_t>(1, (size_t)lu.mpad * lu.mpad));
      KIN_HIP(hipMemsetAsync(gjbuf.p, 0, (size_t)K * std::max<size_t>(1, (size_t)lu.mpad * lu.mpad) * sizeof(double), s));
      // padding entries of the value-ordered stages and the `zero` operand are never written by a factorisation
      KIN_HIP(hipMemsetAsync(Wbuf.p, 0, (size_t)K * slots * (size_t)lu.w_size * sizeof(double), s));
      d_traj.alloc(K); d_res.alloc(K); d_u0.alloc((size_t)K * N);
      K_cap = K; n_slots = slots;
    }
    d_solt.alloc((size_t)K * (size_t)sol_rows);
    if (own_sol) {
      d_sol.alloc((size_t)K * (size_t)sol_rows * N);
      // rows beyond a member's n_saved (a member that failed early) read as zeros in out_u
      KIN_HIP(hipMemsetAsync(d_sol.p, 0, (size_t)K * (size_t)sol_rows * N * sizeof(double), s));
    }
    h_traj.assign(K, ResTrajDev{});
    for (int t = 0; t < K; t++) {
      ResTrajDev& q = h_traj[t];
      double* w = work.p + (size_t)t * per_traj;
      q.u0 = d_u0.p + (size_t)t * N;
      q.k = w; w += al(R);
      q.D = w; w += al((size_t)RES_D_ROWS * N);
      q.y = w; w += al(N); q.psi = w; w += al(N); q.d = w; w += al(N); q.scale = w; w += al(N);
      q.f0 = w; w += al(N); q.f1 = w; w += al(N); q.ytmp = w; w += al(N); q.chunk_start = w; w += al(N);
      q.jv = w; w += al((size_t)H.nnz());
      q.rate = w; w += al(R);
      q.dr = w;
      q.gj_scratch = gjbuf.p + (size_t)t * (size_t)lu.mpad * lu.mpad;
      q.W = Wbuf.p + (size_t)t * slots * (size_t)lu.w_size;
      q.jd = jdbuf.p + (size_t)t * slots * N;
      q.sol = own_sol ? d_sol.p + (size_t)t * (size_t)sol_rows * N : nullptr;
      q.sol_t = d_solt.p + (size_t)t * (size_t)sol_rows;
      q.result = d_res.p + t;
    }
  }
};

void ResidentDeleter::operator()(ResidentSolver* p) const { delete p; }

namespace {

int resident_max_n() {
  static const int v = getenv("KIN_RESIDENT_MAX_N") ? atoi(getenv("KIN_RESIDENT_MAX_N")) : 400;   // (plus RES_MAX_DENSE on the Schur block)
  return v;
}

// ... and on the dense Schur block of the Newton matrix: the in-workgroup dense inverse and GEMV grow with m^3 / m^2 while the
// host-driven path's costs are flat; at the crossover network size of the synthetic CRNs (400 species) m is 130
int resident_max_dense_single() {
  static const int v = getenv("KIN_RESIDENT_MAX_DENSE") ? atoi(getenv("KIN_RESIDENT_MAX_DENSE")) : 160;
  return v;
}

ResidentSolver* get_resident(kin_network* h) {
  if (!h->resident) h->resident.reset(new ResidentSolver(h));
  return h->resident.get();
}

void stats_from(const ResidentSolver& RS, const ResResult& r, int slots, double wall, kin_stats* st) {
  if (!st) return;
  *st = kin_stats{};
  st->n_steps = r.st.n_steps; st->n_rejected = r.st.n_rejected; st->n_rhs = r.st.n_rhs; st->n_jac = r.st.n_jac;
  st->n_factor = r.st.n_factor; st->n_linsolve = r.st.n_linsolve; st->n_newton_fail = r.st.n_newton_fail;
  st->n_chunks = r.st.n_chunks; st->n_restarts = r.st.n_restarts; st->n_retries = r.st.n_retries;
  st->final_abstol = r.final_abstol; st->final_reltol = r.final_reltol; st->wall_seconds = wall;
  st->lu_dense_dim = RS.lu.m; st->lu_sparse_rows = RS.lu.ns; st->lu_rounds = RS.lu.nrounds;
  st->lu_nnz = 2 * RS.lu.nnzU + RS.lu.ns + (int64_t)RS.lu.m * RS.lu.m;
  st->n_lu_reused = r.st.n_lu_reused; st->lu_slots = slots; st->n_bad_pivot = r.st.n_bad_pivot; st->n_lu_dropped = r.st.n_lu_dropped;
}

// common part: parameters, tables of the variable conditions, launch, results
void run_resident(kin_network* h, ResidentSolver& RS, const kin_params& p, const ResGrid& g, int K, int slots, const double* tstops,
                  const double* T_stops, const double* k_table, int64_t n_stops, std::vector<ResResult>& res) {
  hipStream_t s = h->stream;
  const int64_t R = h->host.R;
  ResParams P{};
  res_fill_params(P, p, g);
  res_default_settings(P, slots);
  P.profile = getenv("KIN_RESIDENT_PROFILE") ? 1 : 0;
  RS.d_save.upload(g.save_local, s);
  P.save_local = RS.d_save.p;
  P.n_stops = (int32_t)n_stops;
  P.rate_mode = n_stops > 0 ? (k_table ? 1 : 2) : 0;
  if (n_stops > 0) {
    RS.d_tstops.upload(tstops, (size_t)n_stops, s);
    P.tstops = RS.d_tstops.p;
    if (k_table) { h->table.upload(k_table, (size_t)n_stops * R, s); h->table_rows = n_stops; RS.hn.k_table = h->table.p; }
    else { RS.d_Tstops.upload(T_stops, (size_t)n_stops, s); RS.hn.T_stops = RS.d_Tstops.p; }
  }
  RS.hn.Ea = h->Ea.p; RS.hn.A = h->A.p; RS.hn.has_kmax = h->has_kmax ? 1 : 0; RS.hn.k_max = h->k_max; RS.hn.t_mult = h->t_mult;
  RS.d_net.upload(&RS.hn, 1, s);
  RS.d_par.upload(&P, 1, s);
  RS.d_traj.upload(RS.h_traj.data(), (size_t)K, s);
  // more members than compute units, and two workgroups fit one compute unit's LDS: the build with half the registers per lane
  // (two co-resident workgroups hide each other's latencies: +20-40 % solves/s at 300 species; a member alone is 10-15 % slower)
  static const size_t static_lds = resident_static_lds();
  bool shared_cu = K > h->n_cu && 2 * (RS.dyn_lds + static_lds) <= (size_t)160 * 1024;
  if (const char* e = getenv("KIN_RESIDENT_SHARED_CU")) shared_cu = atoi(e) != 0 && 2 * (RS.dyn_lds + static_lds) <= (size_t)160 * 1024;
  if (shared_cu) launch_resident_shared_cu(K, RS.dyn_lds, RS.d_net.p, RS.d_traj.p, RS.d_par.p, s);
  else launch_resident(K, RS.dyn_lds, RS.d_net.p, RS.d_traj.p, RS.d_par.p, s);
  res.resize(K);
  RS.d_res.download(res.data(), (size_t)K, s);
  KIN_HIP(hipStreamSynchronize(s));
}

int clamp_slots(const ResidentSolver& RS, int K) {
  // LU-cache slots per trajectory: up to RES_MAX_SLOTS, bounded by KIN_LU_CACHE_MB (default 32768) over all trajectories
  size_t budget_mb = 32768;
  if (const char* e = getenv("KIN_LU_CACHE_MB")) budget_mb = (size_t)std::max(1, atoi(e));
  const size_t fit = std::max<size_t>(1, budget_mb * 1024 * 1024 / std::max<size_t>(1, RS.slot_bytes() * (size_t)K));
  int want = RES_MAX_SLOTS;
  if (const char* e = getenv("KIN_LU_CACHE_SLOTS")) want = std::max(1, atoi(e));
  return (int)std::min<size_t>((size_t)std::min(want, RES_MAX_SLOTS), fit);
}

}  // namespace

bool resident_eligible(kin_network* h, const kin_params& p, bool continuous, bool explicit_solver) {
  if (const char* e = getenv("KIN_RESIDENT")) { if (atoi(e) == 0) return false; }
  if (continuous || explicit_solver) return false;
  if (h->host.N > resident_max_n()) return false;
  if (!res_has_grid(p)) return false;
  if (getenv("KIN_TRACE_CHUNK") || getenv("KIN_INJECT_BAD_PIVOT")) return false;
  ResidentSolver* RS = get_resident(h);
  return RS->ok && RS->lu.m <= resident_max_dense_single();
}

// the dynamic LDS is at least 8 (4 N + R) bytes whatever the factorisation looks like (resident_dyn_lds): a network beyond that
// never fits, and is told so WITHOUT the symbolic analysis a ResidentSolver starts with (0.2-1.4 s and a second set of LU plans
// on a 10k-50k species handle)
static bool resident_can_fit(const kin_network* h) {
  return (size_t)(4 * h->host.N + h->host.R) * sizeof(double) <= RES_LDS_BUDGET;
}

bool resident_fits(kin_network* h) { return resident_can_fit(h) && get_resident(h)->ok; }

// Does an ensemble of K members take the one-launch form? Not beyond the kernel's limits, and not FEW members of a network at the
// upper end of what the kernel takes: one workgroup per member lasts as long as its slowest member's resident solve, which from
// ~700 species on is slower than the member's own kin_solve on the host-driven path (table DESIGN 0 "resident vs host-driven"),
// so up to 32 members of such a network are faster as kin_solve calls on host threads / lockstep rounds
// (profiles/r04_ensemble_route_crossover.jsonl: 1 000 species, 16 members 0.5 s against 0.34-1.1 s; from 64 members on the launch
// is 3-4x ahead, at <= 700 species at every K).
bool resident_ensemble_route(kin_network* h, int64_t K) {
  if (!resident_can_fit(h)) return false;
  if (h->host.N > 700 && K < 32) return false;
  return get_resident(h)->ok;
}

int resident_solve(kin_network* h, const kin_params& p, const double* u0, const double* tstops, const double* T_stops,
                   const double* k_table, int64_t n_stops, kin_stats* stats) {
  auto wall0 = std::chrono::steady_clock::now();
  ResidentSolver& RS = *get_resident(h);
  hipStream_t s = h->stream;
  const int64_t N = h->host.N, R = h->host.R;
  const ResGrid g = make_res_grid(p);
  const int slots = clamp_slots(RS, 1);
  // the solution goes straight into the handle's buffer (kin_solution_copy / _max / _dot read it there)
  h->d_sol_u.alloc((size_t)g.cap * N);
  RS.ensure(1, slots, g.cap, false);
  RS.h_traj[0].sol = h->d_sol_u.p;
  RS.d_u0.upload(u0, (size_t)N, s);
  if (n_stops == 0) KIN_HIP(hipMemcpyAsync(RS.h_traj[0].k, h->k.p, (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s));
  std::vector<ResResult> res;
  run_resident(h, RS, p, g, 1, slots, tstops, T_stops, k_table, n_stops, res);
  const ResResult& r = res[0];
  h->n_saved = std::min<int64_t>(r.n_saved, g.cap);
  h->sol_t.resize((size_t)h->n_saved);
  if (h->n_saved > 0) RS.d_solt.download(h->sol_t.data(), (size_t)h->n_saved, s);
  // the rates in force at the end of the solve are what the handle holds afterwards, as on the host-driven path
  if (n_stops > 0) { KIN_HIP(hipMemcpyAsync(h->k.p, RS.h_traj[0].k, (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s)); h->has_rates = true; h->k_pending = false; }
  KIN_HIP(hipStreamSynchronize(s));
  stats_from(RS, r, slots, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count(), stats);
  if (getenv("KIN_RESIDENT_PROFILE")) {
    static const char* names[20] = {"kernel", "factor", "(of which dense inverse)", "corrector attempts", "(solve)", "predict", "change_D",
                                    "accept", "jacobian", "rhs", "(rates + residual)", "(update + sums)", "((rates))", "((stage A))", "((gemv))",
                                    "((stage C))", "((reduce))", "(((reduce: lane sums)))", "(((reduce: first barrier)))", "controller between phases (a worker wavefront waiting for its next command)"};
    fprintf(stderr, "[resident] N=%lld m=%d slots=%d steps=%lld factor=%lld linsolve=%lld wall %.4f s\n", (long long)N, RS.lu.m, slots,
            (long long)r.st.n_steps, (long long)r.st.n_factor, (long long)r.st.n_linsolve,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count());
    for (int i = 0; i < 20; i++) fprintf(stderr, "[resident]   %-26s %9.3f ms\n", names[i], (double)r.prof[i] * 1e-5);
  }
  return r.retcode;
}

// K independent trajectories of one network in ONE launch (one workgroup each): u0[K][N]; rate constants k[K][R], or
// temperatures T[K] (Arrhenius on the device), or the handle's current rates for all; optional discrete rate updates shared by
// the ensemble (tstops + T_stops or k_table). Outputs: out_t[cap], out_u[K][cap][N] (cap = rows of the save grid),
// n_saved[K], retcodes[K], stats[K].
void resident_ensemble(kin_network* h, const kin_params& p, int64_t K, const double* u0, const double* k, const double* T,
                       const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* out_rows,
                       double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats) {
  auto wall0 = std::chrono::steady_clock::now();
  ResidentSolver& RS = *get_resident(h);
  if (!RS.ok) throw KinError(ERR_UNSUPPORTED, "network does not fit the resident integrator: " + RS.why);
  hipStream_t s = h->stream;
  const int64_t N = h->host.N, R = h->host.R;
  const ResGrid g = make_res_grid(p);
  if (out_rows) *out_rows = g.cap;
  const int slots = clamp_slots(RS, (int)K);
  RS.ensure((int)K, slots, g.cap, true);
  RS.d_u0.upload(u0, (size_t)K * N, s);
  if (n_stops == 0) {
    for (int64_t t = 0; t < K; t++) {
      if (k) KIN_HIP(hipMemcpyAsync(RS.h_traj[t].k, k + t * R, (size_t)R * sizeof(double), hipMemcpyHostToDevice, s));
      else if (T) launch_arrhenius(R, h->Ea.p, h->A.p, h->has_kmax, h->k_max, h->t_mult, T[t], RS.h_traj[t].k, s);
      else KIN_HIP(hipMemcpyAsync(RS.h_traj[t].k, h->k.p, (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
  }
  std::vector<ResResult> res;
  run_resident(h, RS, p, g, (int)K, slots, tstops, T_stops, k_table, n_stops, res);
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
  if (out_u) RS.d_sol.download(out_u, (size_t)K * (size_t)g.cap * N, s);
  if (out_t) {   // the save times are the same for every member: taken from the one that got furthest
    int64_t best = 0;
    for (int64_t t = 1; t < K; t++) if (res[t].n_saved > res[best].n_saved) best = t;
    KIN_HIP(hipMemcpyAsync(out_t, RS.d_solt.p + (size_t)best * (size_t)g.cap, (size_t)g.cap * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  KIN_HIP(hipStreamSynchronize(s));
  for (int64_t t = 0; t < K; t++) {
    if (n_saved) n_saved[t] = std::min<int64_t>(res[t].n_saved, g.cap);
    if (retcodes) retcodes[t] = res[t].retcode;
    if (stats) stats_from(RS, res[t], slots, wall, stats + t);
  }
}

}  // namespace kin
