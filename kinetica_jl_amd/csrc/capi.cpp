// extern "C" boundary of libkinetica_hip.so (declarations + reference citations:
// include/kinetica_hip.h). Every entry point converts exceptions into status codes.
#include "../../include/kinetica_hip.h"

#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <cstring>
#include <thread>
#include <vector>

#include "handle.hpp"
#include "lu.hpp"
#include "resident_setup.hpp"
#include "solver.hpp"
#include "solver_kernels.hpp"

using namespace kin;

static thread_local std::string g_create_err;

void kin_network::flush_pending_T(hipStream_t s) {
  if (!k_pending) return;
  launch_arrhenius(host.R, Ea.p, A.p, has_kmax, k_max, t_mult, T_pending, k.p, s);
  k_pending = false;
}

void kin_network::rhs_dev(const double* d_u, double* d_du) {
  if (k_pending) { launch_rates_T(host.R, pending_at(), k.p, d_u, x0.p, x1.p, rate.p, stream); k_pending = false; }
  else launch_rates(host.R, k.p, d_u, x0.p, x1.p, rate.p, stream);
  launch_segsum(rhs_plan.view(), SEG_COEF_SET, rate.p, d_du, SegExtra{}, stream);
}

void kin_network::sweep_dev(int64_t B, const double* d_u, const double* d_k, double* d_du, hipStream_t s) {
  if (!d_k) flush_pending_T(s);   // the handle's own k is about to be read: a temperature still pending is formed first
  if (host.N >= 65535) {
    // the packed sweep records hold species ids in 16 bits; wider networks take the single-state kernels
    // (32-bit ids) state after state on the same stream - correct at any size, one state per launch pair
    for (int64_t b = 0; b < B; b++) {
      launch_rates(host.R, d_k ? d_k + b * host.R : k.p, d_u + b * host.N, x0.p, x1.p, rate.p, s);
      launch_segsum(rhs_plan.view(), SEG_COEF_SET, rate.p, d_du + b * host.N, SegExtra{}, s);
    }
  } else if (host.big_H > 0) {
    big_scratch.alloc((size_t)std::min<int64_t>(B, n_cu) * (size_t)(host.N - host.big_H + host.n_pairs()));
    launch_sweep_big(n_cu, host.N, host.R, host.n_pairs(), B, host.pairs_adjacent, host.big_H, (int32_t)host.big_tail_ptr.size() - 1,
                     big_rec8.p, big_rec.p, big_expl.p, (int32_t)host.big_expl.size(), sweep_k.p, big_spec.p, big_tptr.p, big_tent.p,
                     big_scratch.p, d_u, d_k, k.p, d_du, host.big_tail_by_species, s);
  } else
    launch_sweep(n_cu, host.N, host.R, host.n_pairs(), B, host.pairs_adjacent, host.pairs_block, sweep_rec.p, sweep_k.p,
                 host.pair_rec64.empty() ? nullptr : sweep_rec64.p, sweep_copy.p, (int)host.sweep_copy_species.size(),
                 host.gen_rec8.empty() ? nullptr : gen_rec8.p, gen_expl.p, (int)host.gen_expl.size(), d_u, d_k, k.p, d_du, s);
}

void kin_network::jac_dev(const double* d_u, double* d_vals) {
  if (k_pending) { launch_drates_T(host.R, pending_at(), k.p, d_u, x0.p, x1.p, dr.p, stream); k_pending = false; }
  else launch_drates(host.R, k.p, d_u, x0.p, x1.p, dr.p, stream);
  launch_segsum(jac_plan.view(), SEG_COEF_SET, dr.p, d_vals, SegExtra{}, stream);
}

// every entry point runs on the handle's own device (the one current at kin_network_create), whatever the calling
// thread's current device is: K handles on K host threads can share one GPU or sit on different ones
#define KIN_TRY(h) try { if (h) KIN_HIP(hipSetDevice((h)->device));
#define KIN_CATCH(h)                                                        \
  }                                                                         \
  catch (const KinError& e) {                                               \
    if (h) (h)->err = e.what(); else g_create_err = e.what();               \
    return e.code;                                                          \
  }                                                                         \
  catch (const std::exception& e) {                                         \
    if (h) (h)->err = e.what(); else g_create_err = e.what();               \
    return KIN_ERR_DEVICE;                                                  \
  }                                                                         \
  return KIN_OK;

static void require(bool c, int code, const char* msg) {
  if (!c) throw KinError(code, msg);
}

extern "C" {

const char* kin_version(void) { return "kinetica-hip 0.5 (gfx950)"; }

int kin_abi_version(void) { return KIN_ABI_VERSION; }

int64_t kin_struct_size(int which) {
  return which == 0 ? (int64_t)sizeof(kin_params) : which == 1 ? (int64_t)sizeof(kin_stats) : -1;
}

int kin_device_count(int* n) {
  if (!n) return KIN_ERR_INVALID_ARG;
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) { *n = 0; return KIN_ERR_DEVICE; }
  *n = c;
  return KIN_OK;
}

int kin_set_device(int device) { return hipSetDevice(device) == hipSuccess ? KIN_OK : KIN_ERR_DEVICE; }

const char* kin_last_error(const kin_network* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int kin_network_create(int64_t n_species, int64_t n_reactions, const int64_t* reac_ptr, const int64_t* reac_idx,
                       const int64_t* reac_sto, const int64_t* prod_ptr, const int64_t* prod_idx,
                       const int64_t* prod_sto, int index_base, kin_network** out) {
  kin_network* h = nullptr;
  if (!out) { g_create_err = "out is null"; return KIN_ERR_INVALID_ARG; }
  *out = nullptr;
  try {
    NetworkHost H = compile_network(n_species, n_reactions, reac_ptr, reac_idx, reac_sto, prod_ptr, prod_idx,
                                    prod_sto, index_base);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
      throw KinError(ERR_DEVICE, "no HIP device available (libkinetica_hip has no CPU fallback)");
    h = new kin_network();
    h->host = std::move(H);
    KIN_HIP(hipGetDevice(&h->device));
    KIN_HIP(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    KIN_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const NetworkHost& N = h->host;
    hipStream_t s = h->stream;
    h->x0.upload(N.x0, s); h->x1.upload(N.x1, s);
    h->sp_ptr.upload(N.sp_ptr, s); h->sp_rxn.upload(N.sp_rxn, s); h->sp_coef.upload(N.sp_coef, s);
    if (N.N < 65535) { h->sweep_rec.upload(N.pair_rec, s); h->sweep_k.upload(N.pair_k, s); }
    if (!N.gen_rec8.empty()) { h->gen_rec8.upload(N.gen_rec8, s); h->gen_expl.upload(N.gen_expl, s); }
    if (!N.pair_rec64.empty()) { h->sweep_rec64.upload(N.pair_rec64, s); h->sweep_copy.upload(N.sweep_copy_species, s); }
    if (N.big_H > 0) {
      h->big_rec.upload(N.big_rec, s); h->big_rec8.upload(N.big_rec8, s); h->big_expl.upload(N.big_expl, s);
      h->big_spec.upload(N.big_spec_of_label, s);
      h->big_tptr.upload(N.big_tail_ptr, s); h->big_tent.upload(N.big_tail_ent, s);
    }
    h->rhs_plan.upload(build_seg_plan(N.N, N.sp_ptr.data(), nullptr, N.sp_rxn.data(), nullptr, N.sp_coef.data(), false), s);
    h->jac_plan.upload(build_seg_plan(N.nnz(), N.jc_ptr.data(), nullptr, N.jc_src.data(), nullptr, N.jc_coef.data(), false), s);
    h->k.alloc(N.R); h->rate.alloc(N.R); h->dr.alloc(2 * N.R + 2);
    h->u.alloc(N.N); h->du.alloc(N.N); h->jvals.alloc(N.nnz());
    KIN_HIP(hipStreamSynchronize(s));
    *out = h;
  } catch (const KinError& e) {
    g_create_err = e.what();
    delete h;
    return e.code;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    delete h;
    return KIN_ERR_DEVICE;
  }
  return KIN_OK;
}

// Symbolic analysis of the Newton-matrix factorisation without a device (sizes only): info[0..11] = ns, m, rounds, nnzU, nnzZ,
// nnzV, nnzLZ, nnzNVU, w_size, fused products, gather-plan entries, gather-plan wavefront tasks
int kin_lu_analyze_host(int64_t n_species, int64_t n_reactions, const int64_t* reac_ptr, const int64_t* reac_idx,
                        const int64_t* reac_sto, const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
                        int index_base, int hub_degree, int max_rounds, int max_tail_degree, int max_degree, int min_round,
                        int64_t* info) {
  try {
    const NetworkHost H = compile_network(n_species, n_reactions, reac_ptr, reac_idx, reac_sto, prod_ptr, prod_idx, prod_sto, index_base);
    SparseLU lu;
    lu.host_only = true;
    LUOptions opt;
    if (hub_degree > 0) opt.hub_degree = hub_degree;
    if (max_rounds > 0) opt.max_rounds = max_rounds;
    if (max_tail_degree > 0) opt.max_tail_degree = max_tail_degree;
    if (max_degree > 0) opt.max_degree = max_degree;
    if (min_round > 0) opt.min_round = min_round;
    lu.analyze((int32_t)H.N, H.j_ptr, H.j_col, opt, nullptr);
    if (info) {
      info[0] = lu.ns; info[1] = lu.m; info[2] = lu.nrounds; info[3] = lu.nnzU; info[4] = lu.nnzZ; info[5] = lu.nnzV;
      info[6] = lu.nnzLZ; info[7] = lu.nnzNVU; info[8] = lu.w_size; info[9] = lu.n_fused_products; info[10] = lu.plan_entries;
      info[11] = lu.plan_tasks;
    }
  } catch (const KinError& e) {
    g_create_err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return KIN_ERR_INVALID_ARG;
  }
  return KIN_OK;
}

int kin_network_destroy(kin_network* h) {
  delete h;
  return KIN_OK;
}

int kin_network_sizes(const kin_network* h, int64_t* n_species, int64_t* n_reactions) {
  if (!h) return KIN_ERR_INVALID_ARG;
  if (n_species) *n_species = h->host.N;
  if (n_reactions) *n_reactions = h->host.R;
  return KIN_OK;
}

int kin_set_rates(kin_network* h, const double* k) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(k != nullptr, ERR_INVALID_ARG, "k is null");
  h->k.upload(k, h->host.R, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  h->has_rates = true;
  h->k_pending = false;
  KIN_CATCH(h)
}

int kin_get_rates(kin_network* h, double* k_out) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(k_out != nullptr, ERR_INVALID_ARG, "k_out is null");
  require(h->has_rates, ERR_STATE, "rates were never set");
  h->flush_pending_T(h->stream);
  h->k.download(k_out, h->host.R, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_set_arrhenius(kin_network* h, const double* Ea, const double* A, double k_max, double t_mult) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(Ea && A, ERR_INVALID_ARG, "Ea / A is null");
  h->Ea.upload(Ea, h->host.R, h->stream);
  h->A.upload(A, h->host.R, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  h->has_kmax = !std::isnan(k_max);
  h->k_max = h->has_kmax ? k_max : 1.0;
  h->t_mult = t_mult;
  h->has_arrhenius = true;
  h->t_par_valid = false;
  KIN_CATCH(h)
}

int kin_rates_at(kin_network* h, double T, double* k_out) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(h->has_arrhenius, ERR_STATE, "Arrhenius parameters were never set");
  launch_arrhenius(h->host.R, h->Ea.p, h->A.p, h->has_kmax, h->k_max, h->t_mult, T, h->k.p, h->stream);
  h->k_pending = false;
  if (k_out) h->k.download(k_out, h->host.R, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  h->has_rates = true;
  KIN_CATCH(h)
}

int kin_arrhenius_eval(const double* Ea, const double* A, int64_t n, double k_max, double t_mult, double T,
                       double* k_out) {
  kin_network* h = nullptr;
  KIN_TRY(h)
  require(Ea && A && k_out && n >= 0, ERR_INVALID_ARG, "bad arguments");
  DevBuf<double> dEa, dA, dk;
  dEa.upload(Ea, n); dA.upload(A, n); dk.alloc(n);
  const bool has = !std::isnan(k_max);
  launch_arrhenius(n, dEa.p, dA.p, has, has ? k_max : 1.0, t_mult, T, dk.p, nullptr);
  dk.download(k_out, n);
  KIN_HIP(hipStreamSynchronize(nullptr));
  KIN_CATCH(h)
}

int kin_rate_table(kin_network* h, const double* T, int64_t n_stops, double* out_table) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(h->has_arrhenius, ERR_STATE, "Arrhenius parameters were never set");
  require(T != nullptr && n_stops >= 0, ERR_INVALID_ARG, "bad arguments");
  h->T_stops.upload(T, n_stops, h->stream);
  h->table.alloc((size_t)n_stops * h->host.R);
  launch_rate_table(h->host.R, n_stops, h->Ea.p, h->A.p, h->has_kmax, h->k_max, h->t_mult, h->T_stops.p, h->table.p, h->stream);
  h->table_rows = n_stops;
  if (out_table) h->table.download(out_table, (size_t)n_stops * h->host.R, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_rhs(kin_network* h, const double* u, double* du) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(u && du, ERR_INVALID_ARG, "u / du is null");
  require(h->has_rates, ERR_STATE, "rates were never set");
  h->u.upload(u, h->host.N, h->stream);
  h->rhs_dev(h->u.p, h->du.p);
  h->du.download(du, h->host.N, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_rhs_batched_dev(kin_network* h, int64_t B, const double* d_u, const double* d_k, double* d_du, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0, ERR_INVALID_ARG, "B must be positive");
  require(d_u && d_du, ERR_INVALID_ARG, "null device buffer");
  require(d_k || h->has_rates, ERR_STATE, "rates were never set and no per-state k given");
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  h->sweep_dev(B, d_u, d_k, d_du, s);
  KIN_CATCH(h)
}

int kin_rhs_batched(kin_network* h, int64_t B, const double* u, const double* k, double* du) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0 && u && du, ERR_INVALID_ARG, "bad arguments");
  require(k || h->has_rates, ERR_STATE, "rates were never set and no per-state k given");
  const int64_t N = h->host.N, R = h->host.R;
  hipStream_t s = h->stream;
  h->b_u.upload(u, (size_t)B * N, s);
  h->b_du.alloc((size_t)B * N);
  if (k) h->b_k.upload(k, (size_t)B * R, s);
  h->sweep_dev(B, h->b_u.p, k ? h->b_k.p : nullptr, h->b_du.p, s);
  h->b_du.download(du, (size_t)B * N, s);
  KIN_HIP(hipStreamSynchronize(s));
  KIN_CATCH(h)
}

int kin_jac_nnz(kin_network* h, int64_t* nnz) {
  if (!h || !nnz) return KIN_ERR_INVALID_ARG;
  *nnz = h->host.nnz();
  return KIN_OK;
}

int kin_jac_pattern(kin_network* h, int64_t* rowptr, int64_t* colidx, int index_base) {
  if (!h || !rowptr || !colidx) return KIN_ERR_INVALID_ARG;
  const NetworkHost& N = h->host;
  for (int64_t i = 0; i <= N.N; i++) rowptr[i] = N.j_ptr[i] + index_base;
  for (int64_t e = 0; e < N.nnz(); e++) colidx[e] = N.j_col[e] + index_base;
  return KIN_OK;
}

int kin_jac_values(kin_network* h, const double* u, double* vals) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(u && vals, ERR_INVALID_ARG, "u / vals is null");
  require(h->has_rates, ERR_STATE, "rates were never set");
  h->u.upload(u, h->host.N, h->stream);
  h->jac_dev(h->u.p, h->jvals.p);
  h->jvals.download(vals, h->host.nnz(), h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_solve(kin_network* h, const kin_params* params, const double* u0, const double* tstops,
              const double* T_stops, const double* k_table, int64_t n_stops, int64_t* n_saved, int32_t* retcode,
              kin_stats* stats) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(params && u0, ERR_INVALID_ARG, "params / u0 is null");
  int rc = solve_entry(h, *params, u0, tstops, T_stops, k_table, n_stops, stats);
  if (n_saved) *n_saved = h->n_saved;
  if (retcode) *retcode = rc;
  if (rc != KIN_RETCODE_SUCCESS) throw KinError(ERR_SOLVE_FAILED, "ODE solution failed.");
  KIN_CATCH(h)
}

namespace {

// A solve-only copy of a handle: the compiled tables again on the device, own stream and work vectors; Solver, symbolic LU and LU
// cache are built by its first solve and stay with it
kin_network* clone_for_solves(kin_network* h) {
  std::unique_ptr<kin_network> c(new kin_network());
  c->host = h->host;
  c->device = h->device; c->n_cu = h->n_cu;
  KIN_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  const NetworkHost& N = c->host;
  hipStream_t s = c->stream;
  c->x0.upload(N.x0, s); c->x1.upload(N.x1, s);
  c->sp_ptr.upload(N.sp_ptr, s); c->sp_rxn.upload(N.sp_rxn, s); c->sp_coef.upload(N.sp_coef, s);
  c->rhs_plan.upload(build_seg_plan(N.N, N.sp_ptr.data(), nullptr, N.sp_rxn.data(), nullptr, N.sp_coef.data(), false), s);
  c->jac_plan.upload(build_seg_plan(N.nnz(), N.jc_ptr.data(), nullptr, N.jc_src.data(), nullptr, N.jc_coef.data(), false), s);
  c->k.alloc(N.R); c->rate.alloc(N.R); c->dr.alloc(2 * N.R + 2);
  c->u.alloc(N.N); c->du.alloc(N.N); c->jvals.alloc(N.nnz());
  KIN_HIP(hipStreamSynchronize(s));
  return c.release();
}

int replica_threads_max();
int64_t replica_members_max() {
  return (int64_t)replica_threads_max();
}
int replica_threads_max() {
  const char* e = getenv("KIN_ENSEMBLE_THREADS");   // (read per call: tests change it)
  return std::max(1, e ? atoi(e) : 12);
}

// A SMALL ensemble of a LARGE network: K independent kin_solve calls on K host threads, one solve-only copy of the handle each.
// Every member is, bit for bit, what kin_solve gives for its inputs. A chain of ~14 small dependent launches per step keeps one
// trajectory at 6.7 solves/s (first 2 chunks of the 10k-species network) and K of them at 11 / 17 / 17 / 16 for K = 2 / 4 / 8 / 12 - the
// dispatch rate of the chip's queues; the lockstep rounds of ensemble.cpp only overtake that from K = 16 on.
void replica_ensemble(kin_network* h, const kin_params& p, int64_t K, const double* u0, const double* k, const double* T,
                      const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* n_rows,
                      double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats) {
  const int64_t N = h->host.N, R = h->host.R;
  const int64_t cap = make_res_grid(p).cap;
  if (n_rows) *n_rows = cap;
  // EMPIRICAL (ROCm 7.2, MI355X; tools/replica_sequence.py): members whose streams are among the first ~4 streams of the
  // process slow each other down when they run together - K = 4: 11.4 solves/s, with three or more streams created BEFORE theirs
  // (and kept alive; destroyed ones do not count) 17.0; K = 8: 13.3 -> 16.7; K = 12: 13.9 -> 16.3. Which thread creates a
  // stream makes no difference. bench.py's concurrent_replicas never saw it: torch's own streams play that role there. Four
  // idle placeholder streams are therefore created once per process in front of the first replica's.
  {
    static std::once_flag once;
    std::call_once(once, [] {
      const int n = getenv("KIN_REPLICA_PLACEHOLDER_STREAMS") ? atoi(getenv("KIN_REPLICA_PLACEHOLDER_STREAMS")) : 4;
      for (int i = 0; i < n; i++) { hipStream_t d = nullptr; if (hipStreamCreateWithFlags(&d, hipStreamNonBlocking) != hipSuccess) (void)hipGetLastError(); }
    });
  }
  // Tn threads, member m on thread m mod Tn (a thread solves its members one after the other on its own replica): K <= the
  // thread limit gives every member its own thread, more members share the threads evenly
  const int64_t per = ceil_div(K, (int64_t)replica_threads_max());
  const int64_t Tn = ceil_div(K, per);
  while ((int64_t)h->replicas.size() < Tn) {
    // LU-cache budget of a replica: its share of 70 % of what the device has FREE now, as if every thread of the limit were to get
    // a replica (a later call with more members then finds the same share left for the replicas it adds; the budget is fixed when a
    // replica's Solver is built, at its first solve)
    size_t free_b = 0, total_b = 0;
    KIN_HIP(hipMemGetInfo(&free_b, &total_b));
    h->replicas.push_back(clone_for_solves(h));
    h->replicas.back()->lu_budget_mb = std::max<size_t>(64, (size_t)((double)free_b * 0.7 / (1024.0 * 1024.0)) / (size_t)replica_threads_max());
  }
  KIN_HIP(hipStreamSynchronize(h->stream));
  std::vector<std::string> errs((size_t)Tn);
  std::vector<std::thread> th;
  // the save times are the members' common grid: those of the member that got furthest (the first of them)
  std::mutex grid_mu;
  int64_t grid_n = -1;
  th.reserve((size_t)Tn);
  std::string spawn_err;
  for (int64_t t = 0; t < Tn && spawn_err.empty(); t++) try {
    th.emplace_back([&, t] {
      try {
        kin_network* c = h->replicas[(size_t)t];
        KIN_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        if (h->has_arrhenius) {
          c->Ea.alloc(R); c->A.alloc(R);
          KIN_HIP(hipMemcpyAsync(c->Ea.p, h->Ea.p, (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s));
          KIN_HIP(hipMemcpyAsync(c->A.p, h->A.p, (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s));
          c->has_arrhenius = true; c->has_kmax = h->has_kmax; c->k_max = h->k_max; c->t_mult = h->t_mult;
        }
        for (int64_t m = t; m < K; m += Tn) {
          if (n_stops == 0) {
            if (k) KIN_HIP(hipMemcpyAsync(c->k.p, k + m * R, (size_t)R * sizeof(double), hipMemcpyHostToDevice, s));
            else if (T) launch_arrhenius(R, c->Ea.p, c->A.p, c->has_kmax, c->k_max, c->t_mult, T[m], c->k.p, s);
            else KIN_HIP(hipMemcpyAsync(c->k.p, h->k.p, (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s));
            c->has_rates = true; c->k_pending = false;
          }
          KIN_HIP(hipStreamSynchronize(s));
          kin_stats st{};
          const int rc = solve_entry(c, p, u0 + m * N, tstops, T_stops, k_table, n_stops, &st);
          if (retcodes) retcodes[m] = rc;
          if (stats) stats[m] = st;
          const int64_t ns = std::min<int64_t>(c->n_saved, cap);
          if (n_saved) n_saved[m] = ns;
          if (out_u && ns > 0) c->d_sol_u.download(out_u + (size_t)m * cap * N, (size_t)ns * N, s);
          if (out_u && ns < cap) std::memset(out_u + ((size_t)m * cap + (size_t)ns) * N, 0, (size_t)(cap - ns) * N * sizeof(double));   // rows a failed member never wrote
          KIN_HIP(hipStreamSynchronize(s));
          if (out_t) {
            std::lock_guard<std::mutex> lock(grid_mu);
            if (ns > grid_n) {
              grid_n = ns;
              for (int64_t i = 0; i < ns && i < (int64_t)c->sol_t.size(); i++) out_t[i] = c->sol_t[(size_t)i];
            }
          }
        }
      } catch (const std::exception& e) { errs[(size_t)t] = e.what(); }
    });
  } catch (const std::exception& e) { spawn_err = e.what(); }   // (the threads that did start are joined below before anything is thrown)
  for (auto& x : th) x.join();
  if (!spawn_err.empty()) throw KinError(ERR_DEVICE, "ensemble: could not start a member thread: " + spawn_err);
  for (auto& e : errs) if (!e.empty()) throw KinError(ERR_DEVICE, "ensemble member failed: " + e);
}

}  // namespace

int kin_solve_ensemble(kin_network* h, const kin_params* params, int64_t K, const double* u0, const double* k, const double* T,
                       const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* n_rows,
                       double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(params && u0 && K >= 1, ERR_INVALID_ARG, "params / u0 is null, or K < 1");
  require(!(k && T), ERR_INVALID_ARG, "give per-member rate constants or per-member temperatures, not both");
  require(!(n_stops > 0 && (k || T)), ERR_INVALID_ARG, "discrete rate updates are shared by the ensemble: no per-member k / T with tstops");
  require(!T || h->has_arrhenius, ERR_STATE, "temperatures given but Arrhenius parameters were never set");
  require(res_has_grid(*params), ERR_INVALID_ARG, "an ensemble solve needs a save grid (solve_chunks or save_interval)");
  validate_solve(h, *params, tstops, T_stops, k_table, n_stops, nullptr, nullptr, 0, !(k || T));
  if (n_rows && !out_u && !n_saved) { *n_rows = make_res_grid(*params).cap; }   // size query
  else {
    if (h->k_pending) h->flush_pending_T(h->stream);
    // networks whose trajectory fits one compute unit: one workgroup per member, one launch (resident.cpp); larger ones:
    // lockstep rounds of batched launches (ensemble.cpp). KIN_ENSEMBLE_BATCHED=1 forces the second form.
    const bool force_batched = getenv("KIN_ENSEMBLE_BATCHED") && atoi(getenv("KIN_ENSEMBLE_BATCHED")) != 0;
    const char* route = getenv("KIN_ENSEMBLE_ROUTE");   // resident | threads | lockstep: A/B runs (tools/ensemble_route_crossover.py)
    if (route && !strcmp(route, "threads"))
      replica_ensemble(h, *params, K, u0, k, T, tstops, T_stops, k_table, n_stops, n_rows, out_t, out_u, n_saved, retcodes, stats);
    else if (route && !strcmp(route, "lockstep"))
      batched_ensemble(h, *params, K, u0, k, T, tstops, T_stops, k_table, n_stops, n_rows, out_t, out_u, n_saved, retcodes, stats);
    else if (!force_batched && resident_ensemble_route(h, K))
      resident_ensemble(h, *params, K, u0, k, T, tstops, T_stops, k_table, n_stops, n_rows, out_t, out_u, n_saved, retcodes, stats);
    // few members of a network too large for a compute unit: kin_solve calls on host threads, one member per thread up to the
    // thread limit (beyond it the lockstep rounds are ahead at 3 000 and 10 000 species: profiles/r04_ensemble_route_crossover.jsonl)
    // ... and ANY number of members when the lockstep form does not take the network's factorisation (it needs the fused solve
    // with a dense Schur block; KIN_LU_FUSED=0 or a network without hubs has none): the threads then take several members each
    else if (!force_batched && (K <= replica_members_max() || !ensemble_batched_supported(h, nullptr)))
      replica_ensemble(h, *params, K, u0, k, T, tstops, T_stops, k_table, n_stops, n_rows, out_t, out_u, n_saved, retcodes, stats);
    else
      batched_ensemble(h, *params, K, u0, k, T, tstops, T_stops, k_table, n_stops, n_rows, out_t, out_u, n_saved, retcodes, stats);
  }
  KIN_CATCH(h)
}

int kin_newton_solve(kin_network* h, double c, const double* u, const double* b, double* x) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(u && b && x, ERR_INVALID_ARG, "null buffer");
  require(h->has_rates, ERR_STATE, "rates were never set");
  newton_solve(h, c, u, b, x);
  KIN_CATCH(h)
}

int kin_solve_explicit(kin_network* h, const kin_params* params, const double* u0, const double* tstops, const double* T_stops,
                       const double* k_table, int64_t n_stops, int64_t* n_saved, int32_t* retcode, kin_stats* stats) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(params && u0, ERR_INVALID_ARG, "params / u0 is null");
  require(n_stops >= 0, ERR_INVALID_ARG, "n_stops < 0");
  int rc = solve_entry(h, *params, u0, tstops, T_stops, k_table, n_stops, stats, nullptr, nullptr, 0, true);
  if (n_saved) *n_saved = h->n_saved;
  if (retcode) *retcode = rc;
  if (rc != KIN_RETCODE_SUCCESS) throw KinError(ERR_SOLVE_FAILED, "ODE solution failed.");
  KIN_CATCH(h)
}

int kin_solve_continuous(kin_network* h, const kin_params* params, const double* u0, const double* t_nodes,
                         const double* T_nodes, int64_t n_nodes, int64_t* n_saved, int32_t* retcode, kin_stats* stats) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(params && u0, ERR_INVALID_ARG, "params / u0 is null");
  int rc = solve_entry(h, *params, u0, nullptr, nullptr, nullptr, 0, stats, t_nodes, T_nodes, n_nodes);
  if (n_saved) *n_saved = h->n_saved;
  if (retcode) *retcode = rc;
  if (rc != KIN_RETCODE_SUCCESS) throw KinError(ERR_SOLVE_FAILED, "ODE solution failed.");
  KIN_CATCH(h)
}

int kin_integrator_init(kin_network* h, const kin_params* params, const double* u0, const double* tstops,
                        const double* T_stops, const double* k_table, int64_t n_stops) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(params && u0, ERR_INVALID_ARG, "params / u0 is null");
  require(n_stops >= 0, ERR_INVALID_ARG, "n_stops < 0");
  integrator_init(h, *params, u0, tstops, T_stops, k_table, n_stops);
  KIN_CATCH(h)
}

int kin_integrator_init_continuous(kin_network* h, const kin_params* params, const double* u0, const double* t_nodes,
                                   const double* T_nodes, int64_t n_nodes) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(params && u0, ERR_INVALID_ARG, "params / u0 is null");
  require(n_nodes >= 2, ERR_INVALID_ARG, "need >= 2 (t, T) nodes");
  integrator_init(h, *params, u0, nullptr, nullptr, nullptr, 0, t_nodes, T_nodes, n_nodes);
  KIN_CATCH(h)
}

int kin_integrator_step(kin_network* h, int64_t max_steps, int64_t* steps_taken) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  const int64_t n = integrator_step(h, max_steps);
  if (steps_taken) *steps_taken = n;
  KIN_CATCH(h)
}

int kin_integrator_state(kin_network* h, double* t, double* u, int32_t* retcode, kin_stats* stats) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  integrator_state(h, t, u, retcode, stats);
  KIN_CATCH(h)
}

int kin_solution_size(const kin_network* h, int64_t* n_saved, int64_t* n_species) {
  if (!h) return KIN_ERR_INVALID_ARG;
  if (n_saved) *n_saved = h->n_saved;
  if (n_species) *n_species = h->host.N;
  return KIN_OK;
}

int kin_solution_copy(const kin_network* hc, double* out_t, double* out_u) {
  kin_network* h = const_cast<kin_network*>(hc);
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(out_t && out_u, ERR_INVALID_ARG, "null output buffer");
  std::copy(h->sol_t.begin(), h->sol_t.begin() + h->n_saved, out_t);
  h->d_sol_u.download(out_u, (size_t)h->n_saved * h->host.N, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_solution_max(const kin_network* hc, double* out_umax) {
  kin_network* h = const_cast<kin_network*>(hc);
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(out_umax != nullptr, ERR_INVALID_ARG, "null output buffer");
  require(h->n_saved > 0, ERR_STATE, "no solution stored");
  solution_max(h, out_umax);
  KIN_CATCH(h)
}

int kin_solution_max_dev(const kin_network* hc, double* d_out) {
  kin_network* h = const_cast<kin_network*>(hc);
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(d_out != nullptr, ERR_INVALID_ARG, "null device buffer");
  require(h->n_saved > 0, ERR_STATE, "no solution stored");
  launch_colmax((int)h->host.N, h->n_saved, h->d_sol_u.p, d_out, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_rate_table_dev(kin_network* h, const double* T, int64_t n_stops, double* d_out) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(h->has_arrhenius, ERR_STATE, "Arrhenius parameters were never set");
  require(T != nullptr && d_out != nullptr && n_stops >= 0, ERR_INVALID_ARG, "bad arguments");
  h->T_stops.upload(T, n_stops, h->stream);
  launch_rate_table(h->host.R, n_stops, h->Ea.p, h->A.p, h->has_kmax, h->k_max, h->t_mult, h->T_stops.p, d_out, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_rhs_block_dev(kin_network* h, int64_t r_lo, int64_t r_hi, const double* d_u, double* d_du, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(d_u && d_du, ERR_INVALID_ARG, "null device buffer");
  require(h->has_rates, ERR_STATE, "rates were never set");
  require(0 <= r_lo && r_lo <= r_hi && r_hi <= h->host.R, ERR_INVALID_ARG, "reaction block out of range");
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  h->flush_pending_T(s);
  // rates of the block, zeros elsewhere; the species-major gather then sums exactly this block's contributions
  KIN_HIP(hipMemsetAsync(h->rate.p, 0, (size_t)h->host.R * sizeof(double), s));
  launch_rates(r_hi - r_lo, h->k.p + r_lo, d_u, h->x0.p + r_lo, h->x1.p + r_lo, h->rate.p + r_lo, s);
  launch_segsum(h->rhs_plan.view(), SEG_COEF_SET, h->rate.p, d_du, SegExtra{}, s);
  KIN_CATCH(h)
}

int kin_solution_dot(const kin_network* hc, const double* w, double* out) {
  kin_network* h = const_cast<kin_network*>(hc);
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(w && out, ERR_INVALID_ARG, "null buffer");
  require(h->n_saved > 0, ERR_STATE, "no solution stored");
  DevBuf<double> dw, dout;
  dw.upload(w, h->host.N, h->stream);
  dout.alloc(h->n_saved);
  launch_rowdot((int)h->host.N, h->n_saved, h->d_sol_u.p, dw.p, dout.p, h->stream);
  dout.download(out, h->n_saved, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_rate_table_rows(kin_network* h, const int64_t* rows, int64_t n_rows, double* out) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(rows && out && n_rows >= 0, ERR_INVALID_ARG, "bad arguments");
  require(h->table_rows > 0, ERR_STATE, "no rate table resident (kin_rate_table / kin_solve with a table first)");
  const int64_t R = h->host.R;
  for (int64_t i = 0; i < n_rows; i++) {
    require(rows[i] >= 0 && rows[i] < h->table_rows, ERR_INVALID_ARG, "row index out of range");
    KIN_HIP(hipMemcpyAsync(out + (size_t)i * R, h->table.p + (size_t)rows[i] * R, R * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

}  // extern "C"
