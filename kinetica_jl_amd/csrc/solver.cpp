// kin_solve: the MI355X replacement for init / solve! / reinit! of the reference's stiff
// integrator together with the orchestration around it:
//   * chunkwise local-time solving + output stitching  (reference src/solving/methods.jl:185-303, 717-865)
//   * complete-timespan solving                        (methods.jl:132-183, 655-714)
//   * discrete rate-constant updates at tstops         (src/solving/solve_utils.jl:435-509)
//   * adaptive_solve! tolerance-tightening retries     (solve_utils.jl:376-424)
//
// Integrator. The reference delegates to a user-supplied SciML algorithm (documented choice:
// Sundials CVODE_BDF + KLU, docs/src/getting-started.md:69) that is not vendored, so the
// algorithm restated here is the published quasi-constant-step variable-order BDF/NDF of
// Shampine & Reichelt ("The MATLAB ODE Suite", SIAM J. Sci. Comput. 18, 1997; orders 1-5,
// backward-difference form, modified Newton with an iteration-count dependent safety factor,
// order selection from the error estimates one order down/up). oracle/bdf.py restates the
// same algorithm on the CPU; both are checked against closed-form and high-accuracy truths.
//
// Control flow lives on the host; every vector operation, the RHS, the Jacobian, the LU and
// the triangular solves are device kernels on one stream. A step attempt enqueues predictor +
// two Newton iterations + error estimate blind (kernels turn into no-ops once the device-side
// convergence flag is set) and synchronises ONCE to read a 96-byte control block.
#include "solver.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <functional>
#include <limits>
#include <thread>

#include "lu.hpp"
#include "solver_kernels.hpp"

namespace kin {

// largest power of ten <= h, by exact IEEE operations only (the same double on the host, on the device and in Python:
// resident_core.hpp res_decade_floor, oracle/cpu_bdf.cpp, oracle/bdf.py); h <= 0 or not finite: h itself
static inline double decade_floor(double h) {
  if (!(h > 0.0) || !std::isfinite(h)) return h;
  double p = 1.0;
  while (p > h) p /= 10.0;
  while (p * 10.0 <= h) p *= 10.0;
  return p;
}

namespace {
constexpr double MIN_FACTOR = 0.2, MAX_FACTOR = 10.0, FIRST_MAX_FACTOR = 1e4;
const double KAPPA[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
constexpr double INF = std::numeric_limits<double>::infinity();

enum StepStatus { STEP_OK = 0, STEP_DT_MIN = 1, STEP_UNSTABLE = 2 };

void compute_R(int order, double factor, double R[6][6]) {
  double M[6][6];
  for (int i = 0; i <= order; i++)
    for (int j = 0; j <= order; j++) M[i][j] = 0.0;
  for (int j = 0; j <= order; j++) M[0][j] = 1.0;
  for (int i = 1; i <= order; i++)
    for (int j = 1; j <= order; j++) M[i][j] = ((double)i - 1.0 - factor * (double)j) / (double)i;
  for (int j = 0; j <= order; j++) {
    double p = 1.0;
    for (int i = 0; i <= order; i++) { p *= M[i][j]; R[i][j] = p; }
  }
}
}  // namespace

struct Solver {
  kin_network* h;
  int N;
  hipStream_t s;
  SparseLU lu;
  SegPlanDev resid_plan;                  // Newton residual written straight into the permuted solve vector
  DevBuf<double> D, y, psi, d, scale, f0, f1, ytmp, jv, umax, red;
  DevBuf<BdfCtrl> ctrl;
  BdfCtrl* hc = nullptr;                  // pinned host mirror: the block of the hand-over last waited for (hc_buf[seq & 1])
  BdfCtrl* hc_buf = nullptr;
  BdfCoef cf;
  // integrator state
  double t = 0, h_abs = 0, atol = 0, rtol = 0, newton_tol = 0, dtmin = 0;
  int order = 1, n_equal = 0;
  bool lu_valid = false, jac_current = false, ban_negatives = false;
  // LU cache (CVODE keeps ONE factorisation while gamma drifts < 30 %; with 288 GB of HBM this solver keeps MANY):
  // factorisations stay resident in slots and are reused - across step-size changes AND across restarts (every chunk
  // start and rate update replays the same ramp of step sizes) - whenever a slot's c_fact is within `lu_band` of the
  // current c = h / alpha_k; the Newton update is then scaled by 2 / (1 + c / c_fact) (CVODE's gamrat correction).
  // A slot is refreshed (Jacobian at the predictor + factorisation at the current c) only when a corrector that used
  // it fails. lu_slots == 1 and lu_band == 0 give the round-1 behaviour (a new factorisation at every change of c).
  int lu_slots = 1, cur_slot = 0;
  double lu_band = 0.0;
  int64_t use_clock = 0;
  double fail_score = 0.0;   // leaky count of rejected attempts (history reset at 3, see reset_history)
  kin_stats st{};
  int64_t iters_left = 0;
  // explicit Dormand-Prince 5(4) mode (kin_solve_explicit; SciPy's RK45 is the oracle): stage array K[7][N],
  // state before the last step (dense output), FSAL derivative in K[0]
  bool explicit_mode = false;
  DevBuf<double> rk_K, rk_yold, rk_ynew;
  double rk_h_last = 0.0;
  // continuous-rate solves: called with the segment-local time of every step attempt BEFORE the
  // corrector runs, re-evaluates the rate constants at the conditions of that time
  std::function<void(double)> pre_attempt;

  explicit Solver(kin_network* hh) : h(hh), N((int)hh->host.N), s(hh->stream) {
    const NetworkHost& H = h->host;
    LUOptions opt;
    // Larger networks: eliminate more of the tail sparsely (more rounds, more fill allowed) - every Gauss-Jordan
    // step moves the whole dense block, so its dimension dominates there (C5: 3870 -> 3126, solve -32 %; C3:
    // 1173 -> 981, -10 %); small networks keep few rounds, their cost is the number of dependent launches
    // (C2: the loose setting is 15 % slower). Measured with tools/c5_lu_params.py and bench.py.
    if (N >= 4000) { opt.max_tail_degree = 32; opt.max_rounds = 16; opt.max_degree = 400; }
    lu.analyze(N, H.j_ptr, H.j_col, opt, s);
    {
      // cache size: KIN_LU_CACHE_SLOTS (default 128), bounded by KIN_LU_CACHE_MB (default 32768) of device memory. A
      // restart replays a ramp of step sizes over up to ~12 decades of c; slots sit >= 35 % apart, i.e. ~8 per decade:
      // the cache must hold the whole ramp (~90 slots), or the cyclic sweep evicts every slot just before its next
      // use (32 slots: the full C4 run made 405 k factorisations; its first second, a narrower ramp, only 446)
      int want = 128;
      double band = 0.35;
      size_t budget_mb = 32768;
      if (const char* e = getenv("KIN_INJECT_BAD_PIVOT")) inject_bad_pivot_at = atoll(e);
      if (const char* e = getenv("KIN_SPECULATE")) speculate = atoi(e) != 0;
      fuse_newton = lu.fused_tri && lu.m > 0 && lu.newton_grid() <= 10;
      if (const char* e = getenv("KIN_FUSE_NEWTON")) fuse_newton = atoi(e) != 0 && lu.fused_tri && lu.m > 0;
      if (const char* e = getenv("KIN_LU_CACHE_SLOTS")) want = std::max(1, atoi(e));
      if (const char* e = getenv("KIN_LU_BAND")) band = atof(e);
      if (const char* e = getenv("KIN_LU_CACHE_MB")) budget_mb = (size_t)std::max(1, atoi(e));
      if (h->lu_budget_mb > 0) budget_mb = std::min(budget_mb, h->lu_budget_mb);   // a replica's share (capi.cpp: replica_ensemble)
      const size_t fit = std::max<size_t>(1, budget_mb * 1024 * 1024 / std::max<size_t>(1, lu.slot_bytes()));
      lu_slots = (int)std::min<size_t>(std::min<size_t>((size_t)want, fit), (size_t)LU_MAX_SLOTS);
      d_jdiag.upload(H.j_diag, s);
      d_drift.alloc(LU_MAX_SLOTS + 1);
      KIN_HIP(hipHostMalloc((void**)&h_drift, (LU_MAX_SLOTS + 1) * sizeof(double), hipHostMallocDefault));
      h_drift[LU_MAX_SLOTS] = 0.0;
      lu_band = lu_slots > 1 ? band : 0.0;
      if (lu_slots == 1 && getenv("KIN_LU_BAND")) lu_band = band;   // single slot with a reuse band: CVODE's own scheme
    }
    std::vector<int32_t> yl(N), ident(N);
    lu.yloc.download(yl.data(), N, s);
    KIN_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) ident[i] = i;
    resid_plan.upload(build_seg_plan(N, H.sp_ptr.data(), yl.data(), H.sp_rxn.data(), nullptr, H.sp_coef.data(), false, ident.data()), s);
    D.alloc((size_t)BDF_D_ROWS * N);
    y.alloc(N); psi.alloc(N); d.alloc(N); scale.alloc(N); f0.alloc(N); f1.alloc(N); ytmp.alloc(N); umax.alloc(N);
    jv.alloc(H.nnz());
    ctrl.alloc(1);
    red.alloc((size_t)bdf_reduce_slot() * std::max(bdf_reduce_blocks(N), (lu.fused_tri && lu.m > 0) ? lu.newton_grid() : 0));
    KIN_HIP(hipMemsetAsync(ctrl.p, 0, sizeof(BdfCtrl), s));
    // the step-end hand-over needs device writes to become visible to the spinning host thread while the stream
    // keeps running: fine-grained (coherent), device-mapped pinned memory
    // (two blocks, used alternately by consecutive hand-overs: with a speculatively enqueued step behind the one the host
    // is reading, the next publication must not land in the block being read)
    KIN_HIP(hipHostMalloc((void**)&hc_buf, 2 * sizeof(BdfCtrl), hipHostMallocCoherent | hipHostMallocMapped));
    hc = hc_buf;
    KIN_HIP(hipHostMalloc((void**)&hseq, sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped));
    *hseq = 0;
    if (hipHostGetDevicePointer((void**)&hc_dev, hc_buf, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&hseq_dev, hseq, 0) != hipSuccess || getenv("KIN_NO_FAST_SYNC")) {
      (void)hipGetLastError();
      fast_sync = false; fast_sync_allowed = false; hc_dev = nullptr; hseq_dev = nullptr;
    }
    cf.gamma[0] = 0.0;
    for (int j = 1; j <= BDF_MAX_ORDER; j++) cf.gamma[j] = cf.gamma[j - 1] + 1.0 / j;
    for (int j = 0; j <= BDF_MAX_ORDER; j++) cf.alpha[j] = (1.0 - KAPPA[j]) * cf.gamma[j];
    for (int j = 0; j <= BDF_MAX_ORDER; j++) cf.error_const[j] = KAPPA[j] * cf.gamma[j] + 1.0 / (j + 1);
    cf.error_const[BDF_MAX_ORDER + 1] = 0.0;
  }
  ~Solver() { if (hc_buf) (void)hipHostFree(hc_buf); if (hseq) (void)hipHostFree(hseq); if (h_drift) (void)hipHostFree(h_drift); }

  void set_tols(double a, double r) {
    atol = a; rtol = r;
    // corrector tolerance in the style of ode15s / CVODE: a fixed fraction of the error weight, not RADAU5's sqrt(rtol); see
    // oracle/bdf.py (set_tols). ode15s uses 0.05, CVODE 0.1 (nlscoef). Here 0.03 at the default tolerances, rising to 0.1
    // where the relative tolerance goes below 3e-9 - bdf_newton_frac() in solver_kernels.hpp has the rule and what it rests on.
    newton_tol = std::max(10.0 * std::numeric_limits<double>::epsilon() / rtol, bdf_newton_frac(rtol));
  }

  // step-end hand-over without a stream synchronisation: the corrector launch that decides the attempt publishes the
  // control block into pinned host memory and bumps `*hseq`; the host spins on it (bounded), else falls back to sync_ctrl()
  unsigned long long* hseq = nullptr;
  BdfCtrl* hc_dev = nullptr;                // device-side address of hc
  unsigned long long* hseq_dev = nullptr;
  unsigned long long seq_no = 0;
  bool fast_sync = true, fast_sync_allowed = true;
  int64_t n_sync_fallbacks = 0;   // step-end hand-overs that timed out and went through copy + stream sync (KIN_TIMING=1)
  int sync_ok_streak = 0;
  double sync_wait_s = 0.0;   // host time spent blocked in sync_ctrl (diagnostic, KIN_TIMING=1)
  BdfCtrl* hc_dev_at(unsigned long long seq) const { return hc_dev ? hc_dev + (seq & 1) : nullptr; }
  void wait_ctrl(unsigned long long want) {
    auto t0 = std::chrono::steady_clock::now();
    hc = hc_buf + (want & 1);
    if (fast_sync) {
      for (unsigned spins = 0;; spins++) {
        // (>=: a speculative batch behind the awaited one may have published its own number by the time the host looks)
        if (*(volatile unsigned long long*)hseq >= want) {
          std::atomic_thread_fence(std::memory_order_acquire);
          sync_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          return;
        }
        // not seen within 50 ms (an attempt incl. a large factorisation takes a few ms at most): the platform does not
        // make device writes to pinned host memory visible while the kernel runs - use the synchronising path from now on
        if ((spins & 1023) == 1023 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 50e-3) {
          fast_sync = false; n_sync_fallbacks++; sync_ok_streak = 0; break;
        }
        // a hand-over takes 10-100 us: spin politely (the sibling hardware thread keeps its issue slots), and once the wait
        // is longer than a plain attempt - a factorisation is in the batch, or K handles share fewer cores - give the core away
        __builtin_ia32_pause();
        if (spins > 20000 && (spins & 63) == 63) std::this_thread::yield();
      }
    }
    sync_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    sync_ctrl();
    // a slow wait may have been a one-off (first-launch code loading, a profiler, a preempted GPU): after 64
    // synchronising hand-overs in a row whose sequence number DID arrive the fast path is tried again
    if (fast_sync_allowed && !fast_sync && *(volatile unsigned long long*)hseq >= want && ++sync_ok_streak >= 64) {
      fast_sync = true; sync_ok_streak = 0;
    }
  }
  int64_t iter_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // corrector iterations executed per converged attempt (diagnostic)
  void sync_ctrl() {
    // (with a speculative step enqueued behind the awaited batch the device block has moved on by the time the stream
    // is idle; the awaited batch's last launch has published its block to hc all the same)
    if (!spec.enq) KIN_HIP(hipMemcpyAsync(hc, ctrl.p, sizeof(BdfCtrl), hipMemcpyDeviceToHost, s));
    auto t0 = std::chrono::steady_clock::now();
    KIN_HIP(hipStreamSynchronize(s));
    sync_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  void rhs(const double* u, double* out) { h->rhs_dev(u, out); st.n_rhs++; }

  void eval_jac(const double* u) { h->jac_dev(u, jv.p); st.n_jac++; lu_valid = false; steps_since_jac = 0; jac_stamp_now = st.n_restarts; }

  // The accept of a step (D <- D + d, new difference rows) is deferred and rides in the next step's predictor launch
  // (bdf_accept_predict_kernel: both are one pass over the same columns of D); so does the copy of the new state into
  // the solution buffer when every step is saved. Everything else that reads D calls flush_accept() first.
  bool accept_pending = false;
  int accept_order = 0;
  double* accept_copy = nullptr;
  void flush_accept() {
    if (!accept_pending) return;
    launch_bdf_accept(N, accept_order, D.p, d.p, accept_copy, s);
    accept_pending = false; accept_copy = nullptr;
  }
  void predict() {
    if (accept_pending) {
      launch_bdf_accept_predict(N, accept_order, order, D.p, cf, atol, rtol, y.p, psi.p, d.p, scale.p, ctrl.p, accept_copy, s);
      accept_pending = false; accept_copy = nullptr;
    } else
      launch_bdf_predict(N, order, D.p, cf, atol, rtol, y.p, psi.p, d.p, scale.p, ctrl.p, s);
  }
  // the state after the last accepted step (D[0]), for readers outside the integrator
  double* state_ptr() { flush_accept(); return D.p; }
  // copy of that state into `dst` (a row of the solution buffer): rides with the pending accept when there is one
  void save_state(double* dst) {
    if (accept_pending && !accept_copy) accept_copy = dst;
    else KIN_HIP(hipMemcpyAsync(dst, state_ptr(), (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
  }

  void change_D(int ord, double factor) {
    flush_accept();
    double R[6][6], U[6][6];
    compute_R(ord, factor, R);
    compute_R(ord, 1.0, U);
    BdfMat ru;
    for (int a = 0; a <= ord; a++)
      for (int b = 0; b <= ord; b++) {
        double v = 0.0;
        for (int q = 0; q <= ord; q++) v += R[a][q] * U[q][b];
        ru.v[a][b] = v;
      }
    launch_bdf_change_D(N, ord, ru, D.p, s);
  }

  // (re)start the integrator at time t0 from the state in y (reinit! semantics: order 1, fresh
  // initial step, fresh Jacobian). Returns false when f(y0) is not finite.
  bool restart(double t0, double t_bound) {
    spec = Spec{};
    flush_accept();
    t = t0;
    st.n_restarts++;
    // the Jacobian of the restart state first (it does not depend on the step size chosen below): the drift test of the
    // LU cache then rides on the first synchronisation of the restart
    int n_checked = 0;
    if (!explicit_mode) {
      eval_jac(y.p);
      jac_current = true;
      if (lu_band > 0.0 && lu_drift_max > 0.0) {
        SlotDriftArgs a;
        n_checked = (int)lu.slots.size();
        for (int i = 0; i < n_checked; i++) { a.jd[i] = lu.slots[i].valid ? lu.slots[i].jd.p : nullptr; a.c[i] = lu.slots[i].c_fact; }
        launch_slot_drift(N, n_checked, jv.p, d_jdiag.p, a, d_drift.p, s);
        KIN_HIP(hipMemcpyAsync(h_drift, d_drift.p, (size_t)n_checked * sizeof(double), hipMemcpyDeviceToHost, s));
      }
    }
    rhs(y.p, f0.p);
    launch_bdf_norms(N, y.p, f0.p, nullptr, atol, rtol, ctrl.p, s);
    sync_ctrl();
    for (int i = 0; i < n_checked; i++)
      if (lu.slots[i].valid && !(h_drift[i] <= lu_drift_max)) { lu.slots[i].valid = false; st.n_lu_dropped++; }
    if (hc->nonfinite) return false;
    const double interval = std::fabs(t_bound - t0);
    if (explicit_mode) {
      // the explicit pair keeps SciPy's select_initial_step (its oracle is SciPy's own RK45): exponent 1 / (error order + 1)
      const double d0 = hc->scratch[0], d1 = hc->scratch[1];
      double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
      h0 = std::min(h0, interval);
      launch_axpy_out(N, y.p, h0, f0.p, ytmp.p, s);
      rhs(ytmp.p, f1.p);
      launch_bdf_norms(N, y.p, f0.p, f1.p, atol, rtol, ctrl.p, s);
      sync_ctrl();
      if (hc->nonfinite) return false;
      const double d2 = hc->scratch[2] / h0;
      const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0 * 1e-3) : std::pow(0.01 / std::max(d1, d2), 0.2);
      h_abs = std::min({100.0 * h0, h1, interval});
    } else {
      // CVODE's initial step (cvode.c: cvHin / cvUpperBoundH0 / cvYddNorm; resident_core.hpp restart() has the description),
      // rounded down to a power of ten: every restart's climb passes through the same step sizes, so the matrices of the
      // previous segment's climb are found in the LU cache again (C3, 100 chunks: 363 -> 269 factorisations in round 4's A/B,
      // profiles/r04_h0_decade_ab.txt - round 4's A/B of the grid as an opt-in; the rms deviations from the truths do not move)
      const double hlb = 100.0 * std::numeric_limits<double>::epsilon() * std::max(std::fabs(t0), std::fabs(t_bound));
      double hub = 0.1 * interval;
      if (hub * hc->scratch[3] > 1.0) hub = 1.0 / hc->scratch[3];
      double hg = std::sqrt(hlb * hub), hnew = hg;
      if (hub >= hlb) {
        for (int count = 1; count <= 4; count++) {
          launch_axpy_out(N, y.p, hg, f0.p, ytmp.p, s);
          rhs(ytmp.p, f1.p);
          launch_bdf_norms(N, y.p, f0.p, f1.p, atol, rtol, ctrl.p, s);
          sync_ctrl();
          if (hc->nonfinite) return false;
          const double ydd = hc->scratch[2] / hg;
          hnew = ydd * hub * hub > 2.0 ? std::sqrt(2.0 / ydd) : std::sqrt(hg * hub);
          if (count == 4) break;
          const double hrat = hnew / hg;
          if (hrat > 0.5 && hrat < 2.0) break;
          if (count > 1 && hrat > 2.0) { hnew = hg; break; }
          hg = hnew;
        }
      }
      double h0 = 0.5 * hnew;
      h0 = std::min(std::max(h0, hlb), hub);
      h_abs = decade_floor(std::min(h0, interval));
    }
    if (explicit_mode) {
      rk_K.alloc((size_t)7 * N); rk_yold.alloc(N); rk_ynew.alloc(N);
      KIN_HIP(hipMemcpyAsync(D.p, y.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));          // D[0] = current state
      KIN_HIP(hipMemcpyAsync(rk_K.p, f0.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));      // FSAL: K[0] = f(y0)
      KIN_HIP(hipMemcpyAsync(rk_yold.p, y.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
      rk_h_last = 0.0;
      rk_fsal_pending = false;
      return true;
    }
    launch_bdf_init_D(N, BDF_D_ROWS, y.p, f0.p, h_abs, D.p, s);
    order = 1;
    n_equal = 0;
    fail_score = 0.0;
    first_selection = true;
    return true;
  }

  // After repeated step failures the interpolated difference history is not trusted any more: drop
  // to order 1 and rebuild it from f at the current state, keeping the (already reduced) step size -
  // CVODE's strategy after MXNEF1 error-test failures (oracle/bdf.py: _reset_history).
  void reset_history() {
    flush_accept();
    KIN_HIP(hipMemcpyAsync(ytmp.p, D.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
    rhs(ytmp.p, f0.p);
    launch_bdf_init_D(N, BDF_D_ROWS, ytmp.p, f0.p, h_abs, D.p, s);
    order = 1;
    n_equal = 0;
    lu_valid = false;
    fail_score = 0.0;
    // three failed attempts in a row: nothing cached is trusted any more either
    for (auto& q : lu.slots) q.valid = false;
    force_jac_refresh = true;
  }

  // Warm continuation at a chunk start with unchanged rates (kin_params.solve_chunks == 2): history, order and step size are
  // kept; the Jacobian is evaluated at the chunk's first state and the LU cache's drift guard runs as at a restart
  void resume_chunk() {
    spec = Spec{};
    t = 0.0;
    st.n_restarts++;
    flush_accept();
    KIN_HIP(hipMemcpyAsync(y.p, D.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
    eval_jac(y.p);
    jac_current = true;
    if (lu_band > 0.0 && lu_drift_max > 0.0) {
      SlotDriftArgs a;
      const int n_checked = (int)lu.slots.size();
      for (int i = 0; i < n_checked; i++) { a.jd[i] = lu.slots[i].valid ? lu.slots[i].jd.p : nullptr; a.c[i] = lu.slots[i].c_fact; }
      launch_slot_drift(N, n_checked, jv.p, d_jdiag.p, a, d_drift.p, s);
      KIN_HIP(hipMemcpyAsync(h_drift, d_drift.p, (size_t)n_checked * sizeof(double), hipMemcpyDeviceToHost, s));
      KIN_HIP(hipStreamSynchronize(s));
      for (int i = 0; i < n_checked; i++)
        if (lu.slots[i].valid && !(h_drift[i] <= lu_drift_max)) { lu.slots[i].valid = false; st.n_lu_dropped++; }
    }
  }

  // CVODE's carried convergence rate: each factorisation remembers the contraction its corrector iterations have shown, and
  // the first iteration of a step is judged with it (solver_kernels.hip)
  // The corrector update folded into the solve's last gather launch (stagec_newton_kernel). One launch less per iteration,
  // but every workgroup of the gather then takes part in the reduction hand-over (partial sums + arrival ticket), and that
  // costs ~0.3 us per workgroup: measured per 20-chunk solve, fused against separate: 300 species 0.101 / 0.110 s, 1 000
  // 0.107-0.110 / 0.116, 2 000 0.139 / 0.138, 5 000 0.204 / 0.174, 10 000 (65 workgroups) 0.572 / 0.449 s per 100 chunks.
  // Default: fused when the fused launch has at most 10 workgroups; KIN_FUSE_NEWTON=1 / 0 forces it on / off.
  bool fuse_newton = false;
  // The remembered rate is trusted for `crate_max_age` accepted steps after it was last measured and never across a
  // restart (new rates, new Jacobian): CVODE resets crate at every linear-solver setup, i.e. at least every 20 steps,
  // with a Jacobian at most 50 steps old. Without a bound a slot whose contraction has degraded is never found out -
  // one-iteration steps measure nothing - and the unconverged iterates pile up in the difference history until the
  // error test collapses the step size (seen on the first segment of the C4 ramp).
  int64_t crate_max_age = 10;
  // is the slot's remembered rate fresh enough for the first-iteration test? (it keeps being carried and updated either way)
  bool crate_fresh(const SparseLU::Slot& q) const {
    return q.crate < 1.0 && q.crate_restart == st.n_restarts && st.n_steps - q.crate_step <= crate_max_age;
  }
  // KIN_CRATE_DYMAX: a first correction larger than this (in error-weight units) always gets a second iteration. If the true
  // contraction is worse than the remembered one - up to the 0.2 that a measurement would still accept - the error left behind is
  // 0.25 dy: 0.2 keeps that at the corrector tolerance. (With 1.0 a static 1 400 K solve of a 1k-species network accepted first
  // corrections of 0.66 units on a stale rate, poisoned its difference history and ended in DtLessThanMin at every tolerance.)
  double crate_dy_max = 0.2;
  int last_iters = 0, last_iter_slot = -1;
  double reuse_rate_max = 0.15;  // KIN_LU_RATE_MAX: slowest contraction accepted from a reused factorisation (0.1: C3 11 % slower, in
                                 // refreshes that were not needed; 0.2: not robust, see set_tols; 0.5: 17 % slower on the C4 ramp)
  // KIN_LU_MAX_AGE: a slot is offered for that many restarts after its Jacobian was evaluated. Unlimited reuse is
  // UNSAFE: a direction that was stiff when the slot was made (c J ~ 1e6) and is not any more (its species consumed) is
  // damped to nothing by the old matrix - the corrections vanish, the corrector "converges" at once, and both the
  // convergence test and the error test (which only see the corrections) are blind to the component being left at its
  // predictor; the full C4 ramp then ran into DtLessThanMin a few hundred restarts later (with 5 restarts it runs
  // through). CVODE bounds the same staleness by re-evaluating J at least every 50 steps.
  int64_t lu_max_age = 50;
  // Drift guard: at every restart the Jacobian is fresh; one kernel compares, for every slot, diag(I - c_s J) as it was when
  // the slot was made with what today's Jacobian gives at the same c_s, and slots whose diagonal moved by more than a factor
  // of 1 + lu_drift_max (either way) are dropped. For mass-action kinetics a column's off-diagonal entries are bounded by its
  // diagonal (a reactant's loss terms), so the diagonal is a sound proxy for "a direction that was stiff in the slot and is not
  // any more" - the unsafe case described above, which is a change by ORDERS of magnitude. Rounds 2-4 dropped at 25 %; round 5
  // measured what that costs (profiles/r05_lu_drift_ab.txt: 157 of the 307 factorisations of the 100-chunk C3 solve re-made
  // matrices the guard had dropped although the corrector - whose contraction is watched anyway, reuse_rate_max - converged
  // with them): a factor of 2 keeps the protection and takes 307 -> 242 factorisations, 0.388 -> 0.346 s, C4's 20 chunks
  // 516 -> 291 and 1.51 -> 1.39 s, deviations from the truths inside the spread of the reuse-band study.
  double lu_drift_max = 1.0;
  DevBuf<int32_t> d_jdiag;
  DevBuf<double> d_drift;
  double* h_drift = nullptr;     // pinned
  bool cache_suspended = false;  // a tolerance retry (adaptive_solve!) runs its chunk without the cache
  int64_t jac_stamp_now = 0;     // restart counter at the last Jacobian evaluation
  bool slot_is_fresh = false;
  // `last`: the batch's last launch publishes the control block even when nothing is decided yet (seq = the number the
  // host waits for; the launch that decides publishes it itself - there is no separate error-estimate launch)
  // `spec_it`: an iteration of a speculatively enqueued step - the carried rate comes from the device's control block (the
  // host does not know yet what the step before leaves there), it counts as fresh (enqueue_speculative checked that it will
  // be whatever that step does)
  void newton_iteration(int it, double c, unsigned long long seq, bool last, bool spec_it = false) {
    const int* skip = &ctrl.p->newton_done;
    SparseLU::Slot& q = lu.slots[cur_slot];
    SegExtra ex;
    ex.psi = psi.p; ex.d = d.p; ex.cscal = c; ex.skip = skip;
    // (forming the rates inside the residual gather - three gathers per entry instead of one, no rate launch - was
    // measured slower: 0.565 against 0.553 s on the C3 solve)
    if (h->k_pending) { launch_rates_skip_T(h->host.R, h->pending_at(), h->k.p, y.p, h->x0.p, h->x1.p, h->rate.p, skip, s); h->k_pending = false; }
    else launch_rates_skip(h->host.R, h->k.p, y.p, h->x0.p, h->x1.p, h->rate.p, skip, s);
    launch_segsum(resid_plan.view(), SEG_COEF_BDF, h->rate.p, q.W.p, ex, s);
    // a factorisation made for another c: the update is scaled by 2 / (1 + c / c_fact)
    // (a slot taken under the absolute rule, outside the ratio band: no scaling - it belongs to the stiff limit, where the
    // matrix is c J, and the absolute rule only admits slots whose matrix and this attempt's are both close to the identity)
    const double upd = (q.c_fact != c && std::fabs(c / q.c_fact - 1.0) <= lu_band) ? 2.0 / (1.0 + c / q.c_fact) : 1.0;
    const double rate_max = (lu_band > 0.0 && !cache_suspended && !slot_is_fresh) ? reuse_rate_max : 1.0;
    const double crate0 = q.crate, tol_first = (spec_it || crate_fresh(q)) ? newton_tol : -1.0;
    const bool from_ctrl = spec_it;
    if (fuse_newton) {
      // the solve's last gather stage and the corrector update in one launch
      NewtonFuse f;
      f.skip = skip; f.N = N; f.m = 0; f.off_x = 0; f.x2_species = nullptr;
      f.scale = scale.p; f.y = y.p; f.d = d.p; f.D = D.p; f.order = order;
      f.upd = upd; f.atol = atol; f.rtol = rtol;
      f.ec = cf.error_const[order]; f.ec_m = order > 1 ? cf.error_const[order - 1] : 0.0; f.ec_p = cf.error_const[order + 1];
      f.iter = it; f.maxit = BDF_NEWTON_MAXITER; f.tol = newton_tol; f.rate_max = rate_max; f.crate0 = crate0;
      f.tol_first = tol_first; f.dy_first_max = crate_dy_max; f.crate_from_ctrl = from_ctrl ? 1 : 0; f.ban_negatives = ban_negatives ? 1 : 0;
      f.ctrl = ctrl.p; f.part = red.p; f.host_ctrl = hc_dev_at(seq); f.host_seq = hseq_dev; f.seq = seq; f.publish_always = last ? 1 : 0;
      lu.solve_newton(cur_slot, f, s);
    } else {
      lu.solve(skip, cur_slot, s);
      launch_bdf_newton(N, it, BDF_NEWTON_MAXITER, newton_tol, lu.xloc.p, q.W.p, scale.p, y.p, d.p, upd, rate_max, crate0, tol_first,
                        crate_dy_max, order, D.p, atol, rtol, cf, ctrl.p, red.p, hc_dev_at(seq), hseq_dev, seq, last, s, from_ctrl, ban_negatives);
    }
  }

  // ---- Speculative enqueue of the next step (KIN_SPECULATE=0 switches it off). The quasi-constant-step BDF keeps step size
  // and order for order + 1 steps, so after most steps the next one is fully known in advance IF this one is accepted: its
  // accept + predictor launch and its first corrector batch are enqueued right behind this step's batch, before the host has
  // seen this step's result. The deciding corrector launch writes `spec_go` (1 = accepted, by the host's own rule); the
  // speculative predictor does nothing unless it reads 1, and then the iterations behind it stay no-ops as well. The host
  // takes the batch up when it enters the next step (same arithmetic, bit-identical results) - the device no longer idles
  // for the hand-over, the host's decision and the launch latency of the next predictor (~8-10 us of ~100 per step).
  bool speculate = true;
  bool hold_speculation = false;   // set by integrator_step for the LAST step of a call: nothing half-run may outlive the call
  struct Spec {
    bool enq = false;      // a speculative batch sits behind the batch being waited for
    bool alive = false;    // ... and the device ran it: step() takes it up instead of enqueueing
    unsigned long long seq = 0;
    double t_new = 0.0, hh = 0.0, c = 0.0;
    int slot = -1, blind = 0, order = 0;
  } spec;
  int64_t n_spec = 0, n_spec_dead = 0;   // speculative batches taken up / enqueued for nothing (KIN_TIMING=1)
  // called right after the blind batch of an attempt that ends at t_new (step size h_abs, order, slot cur_slot) is enqueued
  void enqueue_speculative(double t_new, double t_bound, int blind) {
    if (!speculate || hold_speculation || !fast_sync || pre_attempt || trace || inject_bad_pivot_at >= 0 || lu_band <= 0.0 || cache_suspended) return;
    if (n_equal >= order) return;                      // the step after this one may change order / step size
    if (iters_left < 1) return;
    const double t2 = t_new + h_abs;
    if (t2 - t_bound > 0.0) return;                    // would be cut at the segment end
    if (h_abs < std::max(dtmin, 10.0 * (std::nextafter(t_new, INF) - t_new))) return;
    const double hh2 = t2 - t_new, c2 = hh2 / cf.alpha[order];
    if (nearest_slot(c2) != cur_slot) return;          // (the slot can only drop out by the rule that also clears spec_go)
    const SparseLU::Slot& q = lu.slots[cur_slot];
    // the carried rate must count as fresh in the next step whether or not this one measures it anew
    if (!(q.crate < 1.0 && q.crate_restart == st.n_restarts && (st.n_steps + 1) - q.crate_step <= crate_max_age)) return;
    spec.enq = true; spec.alive = false;
    spec.t_new = t2; spec.hh = hh2; spec.c = c2; spec.slot = cur_slot; spec.blind = blind; spec.order = order;
    spec.seq = ++seq_no;
    launch_bdf_accept_predict(N, order, order, D.p, cf, atol, rtol, y.p, psi.p, d.p, scale.p, ctrl.p, nullptr, s, &ctrl.p->spec_go);
    const bool keep = slot_is_fresh;
    slot_is_fresh = q.c_fact == c2;
    for (int b = 0; b < blind; b++) newton_iteration(b, c2, spec.seq, b == blind - 1, true);
    slot_is_fresh = keep;
  }
  void drop_speculation() { if (spec.enq || spec.alive) n_spec_dead++; spec.enq = false; spec.alive = false; }

  void invalidate_lu_keep_counters() {
    for (auto& q : lu.slots) q.valid = false;
    lu_valid = false; cur_slot = 0; force_fresh_lu = false;
  }
  void invalidate_lu() {
    for (auto& q : lu.slots) { q.valid = false; q.c_fact = 0.0; q.last_use = 0; }
    lu_valid = false; cur_slot = 0; use_clock = 0; attempt_no = 0; force_fresh_lu = false; steps_since_jac = 0; cache_suspended = false;
  }
  // slot whose c_fact is closest (in ratio) to c and within the band; -1: none
  int nearest_slot(double c) const {
    int best = -1;
    double bd = 1e300;
    for (int i = 0; i < (int)lu.slots.size(); i++) {
      const SparseLU::Slot& q = lu.slots[i];
      if (!q.valid || st.n_restarts - q.jac_stamp > lu_max_age) continue;
      // continuous rate updates: k(t) moves inside a segment and no restart re-validates the slots - the Jacobian behind
      // a slot may be at most 50 accepted steps old (CVODE's bound on the age of its Jacobian)
      if (pre_attempt && st.n_steps - q.step_stamp > 50) continue;
      const double r = std::fabs(std::log(c / q.c_fact));
      const bool ok = std::fabs(c / q.c_fact - 1.0) <= lu_band;
      if (r < bd && ok) { bd = r; best = i; }
    }
    return best;
  }
  // a slot for a new factorisation: an unused one (allocated on demand), else the least recently used
  int victim_slot() {
    for (int i = 0; i < (int)lu.slots.size(); i++)
      if (!lu.slots[i].valid || st.n_restarts - lu.slots[i].jac_stamp > lu_max_age) return i;
    if ((int)lu.slots.size() < lu_slots) { lu.ensure_slots((int)lu.slots.size() + 1, s); return (int)lu.slots.size() - 1; }
    int v = 0;
    for (int i = 1; i < (int)lu.slots.size(); i++) if (lu.slots[i].last_use < lu.slots[v].last_use) v = i;
    return v;
  }
  // ---- a convergence failure (CVODE: cvHandleNFlag / cvSetEta): a failed corrector cuts the step by ETACF = 0.25 (0.5 until round
  // 3) and does not count towards the history reset (CVODE rebuilds its history only after repeated ERROR-TEST failures: after a
  // reset the order-1 predictor is explicit Euler, which at the step sizes of a late, slowly varying solution is far outside the
  // corrector's convergence region - the complete-timespan C3 solve then halved its step ten more times, each with a
  // factorisation of its own; docs/DESIGN_HISTORY.md R4 has the A/B and the variants that were measured and dropped: a growth
  // cap after a failure, a soft step-size ceiling)
  static constexpr double cf_eta = 0.25;
  bool force_jac_refresh = false;   // a vanished pivot: the next attempt starts from a Jacobian at its own predictor
  // After an error-test rejection the retry gets a factorisation made for its own c (CVODE's rule: a failed error test
  // forces a linear-solver setup): a converged corrector whose matrix was reused carries an iteration error that the
  // rate estimate may understate, and a rejected step moves on to ever older slots at smaller c - without this rule the
  // full C4 ramp spiralled into DtLessThanMin once the cache held the whole step-size ramp.
  bool force_fresh_lu = false;
  int64_t steps_since_jac = 0;
  // fault injection for the tests (KIN_INJECT_BAD_PIVOT=n): the n-th step attempt of a solve finds the flag raised, as
  // if its factorisation had met a vanishing pivot (in accuracy-controlled integration of mass-action kinetics that
  // needs c J_ii ~ 1 on an autocatalytic species and practically never happens by itself)
  int64_t inject_bad_pivot_at = -1, attempt_no = 0;
  bool trace = false;   // KIN_TRACE_CHUNK=n: one line per corrector attempt of chunk n (diagnostic)
  void factor_into(int slot, double c) {
    lu.factor(c, jv.p, slot, &ctrl.p->lu_bad, s);
    lu.slots[slot].last_use = ++use_clock;
    lu.slots[slot].jac_stamp = jac_stamp_now;
    lu.slots[slot].step_stamp = st.n_steps - steps_since_jac;
    if (lu_band > 0.0) {
      lu.slots[slot].jd.alloc(N);
      launch_jac_diag(N, jv.p, d_jdiag.p, lu.slots[slot].jd.p, s);
    }
    cur_slot = slot;
    st.n_factor++;
  }

  // One accepted explicit step (Dormand & Prince 1980, the RK5(4)7M pair; step-size control as in SciPy's
  // RK45: SAFETY 0.9, factors in [0.2, 10], no growth right after a rejection, FSAL). The state lives in D[0].
  StepStatus rk_step(double t_bound) {
    static const double A[6][5] = {{0, 0, 0, 0, 0},
                                   {1.0 / 5, 0, 0, 0, 0},
                                   {3.0 / 40, 9.0 / 40, 0, 0, 0},
                                   {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                                   {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                                   {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
    static const double Bw[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
    static const double E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
    bool rejected = false;
    for (;;) {
      if (iters_left-- <= 0) return STEP_OK;   // caller checks iters_left < 0 -> MaxIters
      const double min_step = std::max(dtmin, 10.0 * (std::nextafter(t, INF) - t));
      if (h_abs < min_step) {
        if (rejected) return STEP_DT_MIN;
        h_abs = min_step;                       // SciPy clips the proposed step to [min_step, max_step] once per step
      }
      double t_new = t + h_abs;
      if (t_new - t_bound > 0.0) t_new = t_bound;
      const double hh = t_new - t;
      h_abs = std::fabs(hh);
      if (pre_attempt) pre_attempt(t_new);
      RkVec w;
      for (int st_i = 1; st_i < 6; st_i++) {
        for (int j = 0; j < 7; j++) w.v[j] = j < st_i ? hh * A[st_i][j] : 0.0;
        launch_rk_combine(N, st_i, w, D.p, rk_K.p, ytmp.p, s);
        rhs(ytmp.p, rk_K.p + (size_t)st_i * N);
      }
      for (int j = 0; j < 7; j++) w.v[j] = j < 6 ? hh * Bw[j] : 0.0;
      launch_rk_combine(N, 6, w, D.p, rk_K.p, rk_ynew.p, s);
      rhs(rk_ynew.p, rk_K.p + (size_t)6 * N);
      RkVec e;
      for (int j = 0; j < 7; j++) e.v[j] = hh * E[j];
      ++seq_no;
      launch_rk_error(N, e, D.p, rk_ynew.p, rk_K.p, atol, rtol, ctrl.p, red.p, hc_dev_at(seq_no), hseq_dev, seq_no, s);
      wait_ctrl(seq_no);
      if (hc->nonfinite) {
        // SciPy would propagate the NaN; here a non-finite stage is treated like a failed step (halve)
        h_abs *= 0.5; rejected = true; st.n_rejected++;
        continue;
      }
      const double err = hc->err_norm;
      if (err < 1.0 && !(ban_negatives && hc->any_negative)) {
        double factor = err == 0.0 ? MAX_FACTOR : std::min(MAX_FACTOR, 0.9 * std::pow(err, -0.2));
        if (rejected) factor = std::min(1.0, factor);
        // accept: y_old <- D[0], D[0] <- y_new, K[0] <- K[6] (FSAL)
        KIN_HIP(hipMemcpyAsync(rk_yold.p, D.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
        KIN_HIP(hipMemcpyAsync(D.p, rk_ynew.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
        rk_h_last = hh;
        h_abs *= factor;
        t = t_new;
        st.n_steps++;
        return STEP_OK;
      }
      h_abs *= (ban_negatives && hc->any_negative && err < 1.0) ? 0.5 : std::max(MIN_FACTOR, 0.9 * std::pow(err, -0.2));
      rejected = true;
      st.n_rejected++;
    }
  }
  // the stage derivatives of the accepted step are needed for dense output until the next step starts:
  // the FSAL copy K[0] <- K[6] is therefore done lazily, at the start of the next step
  bool rk_fsal_pending = false;

  // one accepted step towards t_bound (internally retries rejected attempts)
  StepStatus step(double t_bound) {
    if (explicit_mode) {
      if (rk_fsal_pending) {
        KIN_HIP(hipMemcpyAsync(rk_K.p, rk_K.p + (size_t)6 * N, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));
        rk_fsal_pending = false;
      }
      const StepStatus r = rk_step(t_bound);
      rk_fsal_pending = true;
      return r;
    }
    bool accepted = false, first_attempt = true;
    double safety = 0.9, err_norm = 0.0, t_new = t;
    // this step's first batch is on the device already (enqueued speculatively behind the previous step's)
    bool take_up = spec.alive;
    spec.alive = false; spec.enq = false;
    while (!accepted) {
      if (iters_left-- <= 0) return STEP_OK;  // caller checks iters_left < 0 -> MaxIters
      const double min_step = std::max(dtmin, 10.0 * (std::nextafter(t, INF) - t));
      if (h_abs < min_step) {
        if (!first_attempt) return STEP_DT_MIN;
        // a step that merely STARTS below the resolution of t is raised to it (it only fails if
        // the corrector / error test push it below again)
        change_D(order, min_step / h_abs);
        h_abs = min_step;
        n_equal = 0;
        lu_valid = false;
      }
      first_attempt = false;
      t_new = t + h_abs;
      if (t_new - t_bound > 0.0) {
        t_new = t_bound;
        change_D(order, std::fabs(t_new - t) / h_abs);
        n_equal = 0;
        lu_valid = false;
      }
      const double hh = t_new - t;
      h_abs = std::fabs(hh);
      const double c = hh / cf.alpha[order];
      // the speculative batch ran for another step than the one the host arrived at (cannot happen by the rules of
      // enqueue_speculative; kept as a recovery, not a throw): the device HAS accepted the previous step, so D is current
      // and nothing is pending - the normal path below re-runs the predictor (which clears the control block) and
      // enqueues a batch of its own; what the speculative iterations left in y / d is rebuilt by that predictor
      if (take_up && !(t_new == spec.t_new && hh == spec.hh && c == spec.c && order == spec.order)) { take_up = false; n_spec_dead++; }
      if (pre_attempt) pre_attempt(t_new);
      bool converged = false;
      if (attempt_no++ == inject_bad_pivot_at) KIN_HIP(hipMemsetAsync(&ctrl.p->lu_bad, 1, sizeof(int), s));
      if (force_jac_refresh) {
        predict();
        eval_jac(y.p);
        jac_current = true;
        force_jac_refresh = false;
      }
      // iteration matrix: the cached factorisation closest to this c, else a new one; `fresh` = made in this attempt
      // from the Jacobian of this attempt's predictor
      bool fresh = false;
      const double lu_band = cache_suspended ? 0.0 : this->lu_band;   // shadows the member for this attempt
      if (lu_band > 0.0) {
        const int hit = nearest_slot(c);
        if (take_up && (hit != spec.slot || force_fresh_lu)) { take_up = false; n_spec_dead++; }   // same recovery
        if (hit >= 0 && !force_fresh_lu) { cur_slot = hit; lu.slots[hit].last_use = ++use_clock; st.n_lu_reused++; }
        else {
          if (force_fresh_lu && !jac_current && steps_since_jac > 20) {   // an old Jacobian is refreshed on the way
            predict();
            eval_jac(y.p);
            jac_current = true;
          }
          factor_into(hit >= 0 ? hit : victim_slot(), c);
          fresh = jac_current;
        }
        force_fresh_lu = false;
      } else if (!lu_valid) {
        factor_into(0, c);
        lu_valid = true;
        fresh = jac_current;
      }
      for (;;) {
        // a matrix made in this attempt for this c counts as fresh even when the Jacobian behind it is a few steps old
        slot_is_fresh = fresh || lu.slots[cur_slot].c_fact == c;
        // Blind depth: two iterations are enqueued ahead of the decision, one when this factorisation converged the previous
        // step in its first iteration (the carried rate makes that the common case; the second launch chain would be
        // six no-ops). Whatever is not decided when the host looks gets up to two more iterations per hand-over.
        int it = 0;
        int blind = (last_iters == 1 && last_iter_slot == cur_slot && crate_fresh(lu.slots[cur_slot])) ? 1 : 2;
        unsigned long long batch_seq;
        if (take_up) {          // predictor and `blind` iterations of this attempt are running already
          take_up = false;
          blind = spec.blind; it = blind; batch_seq = spec.seq;
          n_spec++;
        } else {
          predict();
          batch_seq = ++seq_no;
          for (int b = 0; b < blind; b++, it++) newton_iteration(it, c, batch_seq, b == blind - 1);
        }
        enqueue_speculative(t_new, t_bound, blind);
        wait_ctrl(batch_seq);
        while (!hc->newton_done && it < BDF_NEWTON_MAXITER) {
          if (spec.enq) {       // undecided: the speculative batch behind has ended itself and left newton_done raised
            KIN_HIP(hipMemsetAsync(&ctrl.p->newton_done, 0, sizeof(int), s));
            drop_speculation();
          }
          const int more = std::min(2, BDF_NEWTON_MAXITER - it);
          ++seq_no;
          for (int b = 0; b < more; b++, it++) newton_iteration(it, c, seq_no, b == more - 1);
          wait_ctrl(seq_no);
        }
        if (spec.enq && !(hc->newton_done && hc->spec_go)) drop_speculation();   // the device did not run it either
        st.n_rhs += hc->n_iter; st.n_linsolve += hc->n_iter;   // iterations the device executed (blind launches behind the decision are no-ops)
        if (hc->n_iter > 1) {        // a rate was measured in this attempt
          SparseLU::Slot& q = lu.slots[cur_slot];
          q.crate = hc->crate; q.crate_step = st.n_steps; q.crate_restart = st.n_restarts;
        }
        last_iters = (hc->newton_done && hc->converged) ? hc->n_iter : 0;
        last_iter_slot = cur_slot;
        converged = hc->newton_done && hc->converged && !hc->nonfinite;
        if (trace)
          fprintf(stderr, "[trace] t=%.6e h=%.3e order=%d c=%.3e slot=%d c_fact=%.3e fresh=%d jac_cur=%d -> done=%d conv=%d iters=%d dy=%.3e err=%.3e nonfinite=%d lu_bad=%d\n",
                  t, h_abs, order, c, cur_slot, lu.slots[cur_slot].c_fact, (int)fresh, (int)jac_current, hc->newton_done, hc->converged,
                  hc->n_iter, hc->dy_norm, hc->err_norm, hc->nonfinite, hc->lu_bad);
        if (hc->lu_bad) {
          // a pivot of the factorisation in hand vanished (static pivoting): whatever the corrector did with it is
          // discarded, the slot is dropped, and the step is retried at half the size from a fresh Jacobian
          KIN_HIP(hipMemsetAsync(&ctrl.p->lu_bad, 0, sizeof(int), s));
          lu.slots[cur_slot].valid = false;
          lu_valid = false;
          force_jac_refresh = true;
          st.n_bad_pivot++;
          converged = false;
          break;
        }
        if (converged) break;
        st.n_newton_fail++;
        if (lu_band > 0.0) {
          // matrix made for this very step from a current Jacobian: the step itself is too long. Otherwise the
          // slot is refreshed: Jacobian at the predictor (if not current), factorisation at this c, one retry
          if (fresh) break;
          if (!jac_current) {
            predict();
            eval_jac(y.p);
            jac_current = true;
          }
          factor_into(cur_slot, c);
          fresh = true;
          continue;
        }
        if (jac_current) break;
        predict();
        eval_jac(y.p);
        jac_current = true;
        factor_into(0, c);
        lu_valid = true;
      }
      if (!converged || (ban_negatives && hc->any_negative)) {
        // a failed corrector cuts the step by cf_eta; isoutofdomain (methods.jl:169-171) halves it and counts like an
        // error-test failure
        const bool conv_failure = !converged;
        const double eta = conv_failure ? cf_eta : 0.5;
        h_abs *= eta;
        change_D(order, eta);
        n_equal = 0;
        lu_valid = false;
        st.n_rejected++;
        first_selection = false;   // (CVODE: any failed attempt sets etamax = 1, the first step's 1e4 is gone)
        if (!conv_failure) fail_score += 1.0;
        if (fail_score >= 3.0 && order > 1) reset_history();
        continue;
      }
      iter_hist[std::min(hc->n_iter, 7)]++;
      // a reused factorisation that needed every allowed iteration is too stale to be offered again
      if (lu_band > 0.0 && !fresh && hc->n_iter >= BDF_NEWTON_MAXITER) lu.slots[cur_slot].valid = false;
      safety = 0.9 * (2.0 * BDF_NEWTON_MAXITER + 1.0) / (2.0 * BDF_NEWTON_MAXITER + hc->n_iter);
      err_norm = hc->err_norm;
      if (err_norm > 1.0) {
        const double factor = std::max(MIN_FACTOR, safety * std::pow(err_norm, -1.0 / (order + 1)));
        h_abs *= factor;
        change_D(order, factor);
        n_equal = 0;
        // without the cache the matrix is kept for the retry (the corrector converged with it; the update is scaled for
        // the new c); with the cache the retry gets a factorisation of its own
        force_fresh_lu = lu_band > 0.0;
        st.n_rejected++;
        first_selection = false;
        fail_score += 1.0;
        if (fail_score >= 3.0 && order > 1) reset_history();
      } else {
        if (hc->any_negative & 2) return STEP_UNSTABLE;   // solver_kernels.hpp BDF_NEG_DEEP: the negative excursion, given up early
        accepted = true;
      }
    }
    st.n_steps++;
    steps_since_jac++;
    fail_score = std::max(0.0, fail_score - 0.2);
    n_equal++;
    t = t_new;
    if (spec.enq) {   // the device has accepted this step and is computing the next one: nothing is pending
      spec.alive = true; spec.enq = false;
      accept_pending = false; accept_copy = nullptr;
    } else {
      accept_pending = true; accept_order = order; accept_copy = nullptr;   // rides in the next predictor launch
    }
    jac_current = false;
    pending_order_change = (n_equal >= order + 1);
    if (pending_order_change) {
      err_m = order > 1 ? hc->err_m_norm : INF;
      err_p = order < BDF_MAX_ORDER ? hc->err_p_norm : INF;
      err_o = err_norm;
      safety_o = safety;
    }
    return STEP_OK;
  }

  // order / step-size selection after an accepted step (done after the dense-output saves,
  // which need the differences of the step just taken)
  bool pending_order_change = false, first_selection = false;
  double err_m = 0, err_p = 0, err_o = 0, safety_o = 0.9;
  void select_order() {
    if (explicit_mode || !pending_order_change) return;
    pending_order_change = false;
    const double norms[3] = {err_m, err_o, err_p};
    double best = -1.0;
    int arg = 1;
    for (int i = 0; i < 3; i++) {
      double f;
      if (norms[i] == 0.0) f = INF;
      else if (std::isinf(norms[i])) f = 0.0;
      else f = std::pow(norms[i], -1.0 / (order + i));
      if (f > best) { best = f; arg = i; }
    }
    order += arg - 1;
    // growth cap 1e4 at the first selection after a (re)initialisation, 10 afterwards (CVODE: ETAMX1, ETAMX2 / ETAMX3)
    double factor = std::min(first_selection ? FIRST_MAX_FACTOR : MAX_FACTOR, safety_o * best);
    first_selection = false;
    h_abs *= factor;
    change_D(order, factor);
    n_equal = 0;
    lu_valid = false;
  }

  // dense output of the step that ended at t (step size h_abs, differences D of `order`)
  void interpolate(double ts, double* out) {
    if (explicit_mode) {
      // SciPy's RkDenseOutput for RK45: y(t) = y_old + h x sum_j K_j (sum_p P[j][p] x^p), x = (t - t_old) / h
      static const double P[7][4] = {
          {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
          {0, 0, 0, 0},
          {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
          {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
          {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
          {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
          {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};
      const double x = (ts - (t - rk_h_last)) / rk_h_last;
      RkVec w;
      for (int j = 0; j < 7; j++) {
        double acc = 0.0, xp = x;
        for (int q = 0; q < 4; q++) { acc += P[j][q] * xp; xp *= x; }
        w.v[j] = rk_h_last * acc;
      }
      launch_rk_combine(N, 7, w, rk_yold.p, rk_K.p, out, s);
      return;
    }
    flush_accept();
    BdfVec p;
    double prod = 1.0;
    for (int j = 0; j < order; j++) {
      prod *= (ts - (t - h_abs * j)) / (h_abs * (1.0 + j));
      p.v[j + 1] = prod;
    }
    launch_bdf_interp(N, order, D.p, p, out, s);
  }
};

// ------------------------------------------------------------------------------------------
// orchestration
// ------------------------------------------------------------------------------------------
namespace {

struct SaveBuf {
  kin_network* h;
  int N;
  int64_t cap = 0;
  void reserve(int64_t rows) {
    if (rows <= cap) return;
    int64_t ncap = std::max<int64_t>(rows, cap * 2);
    DevBuf<double> nb;
    nb.alloc((size_t)ncap * N);
    if (h->n_saved > 0)
      KIN_HIP(hipMemcpyAsync(nb.p, h->d_sol_u.p, (size_t)h->n_saved * N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    KIN_HIP(hipStreamSynchronize(h->stream));
    std::swap(h->d_sol_u.p, nb.p);
    std::swap(h->d_sol_u.n, nb.n);
    cap = ncap;
  }
  double* row(int64_t i) { return h->d_sol_u.p + (size_t)i * N; }
  void push_time(double t) { h->sol_t.push_back(t); h->n_saved++; }
};

// dtmin handed to the integrator: the caller's value, else what the reference passes - eps(solve_chunkstep) for
// chunkwise solves (methods.jl:232, 770), eps(tspan[end]) for complete-timespan ones (methods.jl:164, 694);
// Julia's eps(x) is the spacing of the doubles at x
double resolve_dtmin(const kin_params& p) {
  if (p.dtmin > 0.0) return p.dtmin;
  const double x = std::fabs(p.solve_chunks != 0 ? p.solve_chunkstep : p.tspan1);
  return std::nextafter(x, INF) - x;
}

// sets the handle's current rates for time-stop index si
void apply_rates(kin_network* h, const double* T_stops, bool have_table, int64_t si) {
  const int64_t R = h->host.R;
  if (have_table) {
    KIN_HIP(hipMemcpyAsync(h->k.p, h->table.p + (size_t)si * R, R * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  } else {
    launch_arrhenius(R, h->Ea.p, h->A.p, h->has_kmax, h->k_max, h->t_mult, T_stops[si], h->k.p, h->stream);
  }
  h->has_rates = true;
  h->k_pending = false;
}

}  // namespace

void validate_solve(kin_network* h, const kin_params& p, const double* tstops, const double* T_stops, const double* k_table,
                    int64_t n_stops, const double* t_nodes, const double* T_nodes, int64_t n_nodes, bool need_handle_rates) {
  // ---- validation (ODESimulationParams constructor, params.jl:77-104)
  if (!(p.tspan0 < p.tspan1)) throw KinError(ERR_INVALID_ARG, "Invalid time span");
  if (!(p.abstol > 0) || !(p.reltol > 0)) throw KinError(ERR_INVALID_ARG, "tolerances must be positive");
  const bool chunks = p.solve_chunks != 0;
  const bool has_save = p.save_interval >= 0;
  if (chunks) {
    if (!(p.solve_chunkstep > 0)) throw KinError(ERR_INVALID_ARG, "solve_chunkstep must be positive");
    const double q = p.tspan1 / p.solve_chunkstep;   // Int(tspan[2] / solve_chunkstep) must be exact (params.jl:89-99)
    if (q != std::floor(q) || q < 1) throw KinError(ERR_INVALID_ARG, "Simulation timespan is not divisible by requested chunkwise simulation step size");
    if (has_save && p.save_interval > p.solve_chunkstep) throw KinError(ERR_INVALID_ARG, "Solution save interval must be less than chunkwise simulation step size");
    if (has_save && !(p.save_interval > 0)) throw KinError(ERR_INVALID_ARG, "save_interval must be positive");
  }
  const bool continuous = n_nodes > 0;
  if (continuous) {
    if (n_stops > 0) throw KinError(ERR_INVALID_ARG, "continuous and discrete rate updates are mutually exclusive");
    if (!t_nodes || !T_nodes || n_nodes < 2) throw KinError(ERR_INVALID_ARG, "need >= 2 (t, T) nodes");
    if (!h->has_arrhenius) throw KinError(ERR_STATE, "continuous rates need the Arrhenius parameters");
    for (int64_t i = 1; i < n_nodes; i++)
      if (!(t_nodes[i] >= t_nodes[i - 1])) throw KinError(ERR_INVALID_ARG, "t_nodes must be non-decreasing");
  }
  const bool variable = n_stops > 0;
  if (variable) {
    if (!tstops) throw KinError(ERR_INVALID_ARG, "tstops is null");
    if (!k_table && !T_stops) throw KinError(ERR_INVALID_ARG, "need k_table or T_stops with tstops");
    if (!k_table && !h->has_arrhenius) throw KinError(ERR_STATE, "T_stops given but Arrhenius parameters were never set");
    for (int64_t i = 1; i < n_stops; i++)
      if (!(tstops[i] > tstops[i - 1])) throw KinError(ERR_INVALID_ARG, "tstops must be strictly increasing");
  } else if (!continuous && need_handle_rates && !h->has_rates) {
    throw KinError(ERR_STATE, "rates were never set");
  }
}

int solve_entry(kin_network* h, const kin_params& p, const double* u0, const double* tstops, const double* T_stops,
                const double* k_table, int64_t n_stops, kin_stats* stats, const double* t_nodes, const double* T_nodes,
                int64_t n_nodes, bool explicit_solver) {
  auto wall0 = std::chrono::steady_clock::now();
  const int64_t N = h->host.N, R = h->host.R;
  validate_solve(h, p, tstops, T_stops, k_table, n_stops, t_nodes, T_nodes, n_nodes, true);
  int64_t n_chunks = 1;
  const bool chunks = p.solve_chunks != 0;
  const bool has_save = p.save_interval >= 0;
  if (chunks) n_chunks = (int64_t)(p.tspan1 / p.solve_chunkstep);
  const bool continuous = n_nodes > 0;
  const bool variable = n_stops > 0;
  // small networks: the whole solve in one launch, one workgroup owns the trajectory (resident.cpp)
  if (resident_eligible(h, p, continuous, explicit_solver)) {
    if (h->k_pending) h->flush_pending_T(h->stream);
    return resident_solve(h, p, u0, tstops, T_stops, k_table, n_stops, stats);
  }
  if (!h->solver) h->solver.reset(new Solver(h));
  Solver& S = *h->solver;
  hipStream_t s = h->stream;
  // whatever way this call ends (a throw included): no temperature stays pending on the handle (kin_rhs / kin_jac and the
  // batched sweeps would otherwise see rate constants of different temperatures), the hook that captures this call's
  // locals is gone, and no speculative batch is left half-taken-up
  struct ExitGuard {
    kin_network* h; Solver& S;
    ~ExitGuard() {
      S.pre_attempt = nullptr;
      S.spec = Solver::Spec{};
      if (h->k_pending) { try { h->flush_pending_T(h->stream); } catch (...) { h->k_pending = false; } }
    }
  } exit_guard{h, S};
  S.st = kin_stats{};
  S.invalidate_lu();   // the LU cache lives within one solve: identical calls give identical results
  S.accept_pending = false; S.accept_copy = nullptr;   // nothing of an earlier call (its solution buffer may be gone)
  S.spec = Solver::Spec{};
  S.explicit_mode = explicit_solver;
  S.sync_wait_s = 0.0;
  std::fill(S.iter_hist, S.iter_hist + 8, 0);
  S.ban_negatives = p.ban_negatives != 0;
  S.dtmin = resolve_dtmin(p);
  double abstol = p.abstol, reltol = p.reltol;
  S.set_tols(abstol, reltol);

  const bool have_table = variable && k_table != nullptr;
  if (have_table) {
    h->table.upload(k_table, (size_t)n_stops * R, s);
    h->table_rows = n_stops;
  }

  // ---- output storage
  h->sol_t.clear();
  h->n_saved = 0;
  SaveBuf sb{h, (int)N};
  // per-chunk local save grid: 0:save_interval:chunkstep (methods.jl:756-758); element i is the
  // correctly rounded i*save_interval as produced by Julia's float ranges
  std::vector<double> save_local;
  double span_len = chunks ? p.solve_chunkstep : (p.tspan1 - p.tspan0);
  if (chunks || has_save) {
    const double si = has_save ? p.save_interval : p.solve_chunkstep;
    const double base = chunks ? 0.0 : p.tspan0;
    const double last = chunks ? p.solve_chunkstep : p.tspan1;
    const int64_t cnt = (int64_t)std::floor(span_len / si + 1e-9) + 1;
    for (int64_t i = 0; i < cnt; i++) save_local.push_back(std::min(base + (double)i * si, last));
    if (!chunks && save_local.back() < last) save_local.push_back(last);  // save_end
    // chunkwise: saveat_local = collect(0:save_interval:chunkstep) is a VECTOR, so the chunk end is saved only when it
    // is a grid point; otherwise the chunk's last saved point lies before the chunk end (methods.jl:756-758, 829-846)
    if (chunks && std::fabs(save_local.back() - last) <= 1e-9 * last) save_local.back() = last;
  }
  const int64_t L = (int64_t)save_local.size();
  // true: the chunk end is the last local save point (the usual case). false (save_interval does not divide the
  // chunk): the final chunk's last output is the dense-output value at its last local save point. (The reference
  // then also restarts every chunk from integ.sol.u[end] = the state at that last SAVED point while labelling it
  // as the chunk end - a defect that is not copied: chunks always continue from the state at the chunk end.)
  const bool save_hits_end = chunks && L > 0 && save_local.back() == p.solve_chunkstep;
  if (chunks) sb.reserve((L - 1) * n_chunks + 1);
  else sb.reserve(has_save ? L : 1024);

  // continuous rates: T(t) = linear interpolation of the profile solution (what the DiffEqArray
  // functor of src/utils.jl:135-139 does), k = calculator(T(t)) re-evaluated on the device
  auto T_of = [&](double tg) {
    if (tg <= t_nodes[0]) return T_nodes[0];
    if (tg >= t_nodes[n_nodes - 1]) return T_nodes[n_nodes - 1];
    const int64_t i = std::upper_bound(t_nodes, t_nodes + n_nodes, tg) - t_nodes;   // t_nodes[i-1] <= tg < t_nodes[i]
    const double dt = t_nodes[i] - t_nodes[i - 1];
    const double th = dt > 0 ? (tg - t_nodes[i - 1]) / dt : 1.0;
    return (1.0 - th) * T_nodes[i - 1] + th * T_nodes[i];
  };
  double seg_origin = 0.0;   // global time of the current segment's tau = 0
  if (continuous) {
    // (no launch: the first kernel of the attempt that reads k forms it from this temperature, handle.hpp)
    S.pre_attempt = [&](double tau) { h->set_pending_T(T_of(seg_origin + tau)); };
    h->has_rates = true;
  } else {
    S.pre_attempt = nullptr;
  }
  // initial state
  S.y.upload(u0, N, s);
  int64_t next_stop = 0;      // first tstop not yet applied
  int retcode = KIN_RETCODE_SUCCESS;
  const double t_origin = chunks ? 0.0 : p.tspan0;
  DevBuf<double> chunk_start;
  chunk_start.alloc(N);
  bool have_history = false, rates_changed = false;
  int64_t rates_in_force = -1;
  // every segment start re-initialises like the reference (reinit!, methods.jl:260, 819), except the chunk starts of
  // kin_params.solve_chunks == 2 whose rates did not change (resume_chunk). Carrying the history across RATE UPDATES was built and
  // measured in round 4 (3 180 rejected steps and 4x the time on the C4 prefix) and is gone (docs/DESIGN_HISTORY.md R4).

  // initial rates = calculator at the initial conditions (methods.jl:672, 734); a tstop at the
  // very start overrides it below
  if (variable && !have_table) { /* rates at T_stops[0] are applied by the loop when tstops[0] == start */ }

  // KIN_PROGRESS=<seconds>: a status line on stderr at most that often (the reference's `progress` option drives a
  // progress bar from the same place, methods.jl:822-827)
  const double progress_every = getenv("KIN_PROGRESS") ? std::max(1.0, atof(getenv("KIN_PROGRESS"))) : 0.0;
  auto progress_last = wall0;
  const long long trace_chunk = getenv("KIN_TRACE_CHUNK") ? atoll(getenv("KIN_TRACE_CHUNK")) : -1;
  for (int64_t nc = 0; nc < n_chunks && retcode == KIN_RETCODE_SUCCESS; nc++) {
    S.st.n_chunks++;
    S.cache_suspended = false;
    S.trace = (nc == trace_chunk);
    if (progress_every > 0.0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - progress_last).count() >= progress_every) {
      progress_last = std::chrono::steady_clock::now();
      fprintf(stderr, "[kin_solve] chunk %lld / %lld, %.1f s, %lld steps, %lld factorisations, %lld restarts\n", (long long)nc,
              (long long)n_chunks, std::chrono::duration<double>(progress_last - wall0).count(), (long long)S.st.n_steps,
              (long long)S.st.n_factor, (long long)S.st.n_restarts);
      fflush(stderr);
    }
    const double t_start_global = chunks ? p.solve_chunkstep * (double)nc : p.tspan0;
    const double t_end_global = chunks ? t_start_global + p.solve_chunkstep : p.tspan1;
    const double shift = chunks ? (double)nc * p.solve_chunkstep : 0.0;   // global = local + shift
    const double t_loc0 = chunks ? 0.0 : p.tspan0;
    const double t_loc1 = chunks ? p.solve_chunkstep : p.tspan1;
    (void)t_origin;
    // local stops of this chunk: tg - nc*chunkstep for tstops in [t_start, t_end) (methods.jl:798-800);
    // the complete-timespan variant takes all of them (methods.jl:697)
    const int64_t stop_first = next_stop;
    KIN_HIP(hipMemcpyAsync(chunk_start.p, S.y.p, N * sizeof(double), hipMemcpyDeviceToDevice, s));
    const int64_t saved_at_chunk_start = h->n_saved;
    const size_t times_at_chunk_start = h->sol_t.size();

    int attempts = 0;
    for (;;) {  // adaptive_solve! (solve_utils.jl:376-424)
      attempts++;
      if (attempts > 1) have_history = false;   // a retry starts cold from the chunk's first state
      retcode = KIN_RETCODE_SUCCESS;
      S.iters_left = p.maxiters;
      int64_t stop_i = stop_first;
      // rates in force at the chunk start: the last stop at or before it (zero-order hold)
      while (variable && stop_i < n_stops && tstops[stop_i] <= t_start_global) {
        stop_i++;
      }
      if (variable) {
        const int64_t want = stop_i > 0 ? stop_i - 1 : 0;   // before the first stop: initial conditions
        if (want != rates_in_force) { apply_rates(h, T_stops, have_table, want); rates_in_force = want; rates_changed = true; }
      }
      int64_t save_i = 0;
      double t_seg = t_loc0;
      bool failed = false;
      // save the chunk's first point (local t = 0)
      auto save_state_now = [&](double t_local) {
        sb.reserve(h->n_saved + 1);
        KIN_HIP(hipMemcpyAsync(sb.row(h->n_saved), S.y.p, N * sizeof(double), hipMemcpyDeviceToDevice, s));
        sb.push_time(t_local + shift);
      };
      if (L > 0) { save_state_now(save_local[0]); save_i = 1; }
      else save_state_now(t_loc0);   // saveat = []: every step, starting with the initial state

      while (t_seg < t_loc1 && !failed) {
        // segment end: next tstop inside this chunk, else the chunk end
        double seg_end = t_loc1;
        bool ends_at_stop = false;
        if (variable && stop_i < n_stops && tstops[stop_i] < t_end_global) {
          const double loc = tstops[stop_i] - shift;
          if (loc < t_loc1) { seg_end = loc; ends_at_stop = true; }
        }
        if (seg_end > t_seg) {
          // Every segment is integrated in segment-local time tau in [0, seg_len]: the rates are
          // constant inside a segment (autonomous system), and restarting at tau = 0 keeps the tiny
          // first steps of a restart (1e-20 s is common) above the floating-point resolution of the
          // time variable - the underflow the reference's chunking exists to avoid
          // (docs/src/development/implementation-details.md:5-28) but re-creates at tstops > 0.
          const double seg_len = seg_end - t_seg;
          seg_origin = t_seg + shift;
          if (continuous) S.pre_attempt(0.0);   // rates at the segment start for f0 / J of the restart
          // kin_params.solve_chunks == 2: warm continuation at a chunk start whose rates did not change (a StaticODESolve's
          // chunk boundaries; between the rate updates of a ramp the integrator is re-initialised as the reference does)
          const bool warm_chunk = p.solve_chunks == 2 && have_history && !rates_changed && !continuous && !S.explicit_mode;
          if (warm_chunk) S.resume_chunk();
          else if (!S.restart(0.0, seg_len)) { retcode = KIN_RETCODE_UNSTABLE; failed = true; break; }
          have_history = true;
          rates_changed = false;
          while (S.t < seg_len) {
            StepStatus ss = S.step(seg_len);
            if (S.iters_left < 0) { retcode = KIN_RETCODE_MAXITERS; failed = true; break; }
            if (ss == STEP_DT_MIN) { retcode = KIN_RETCODE_DTLESSTHANMIN; failed = true; break; }
            if (ss == STEP_UNSTABLE) { retcode = KIN_RETCODE_UNSTABLE; failed = true; break; }
            const double t_abs = S.t >= seg_len ? seg_end : t_seg + S.t;   // chunk-local time reached
            // saves covered by this step
            if (L > 0) {
              // the chunk's last local point: dropped except on the final chunk (methods.jl:829-846), where it is the
              // chunk's end state (saved below) or, off the grid, a dense-output value
              const int64_t last = (chunks && !(nc == n_chunks - 1 && !save_hits_end)) ? L - 1 : L;
              while (save_i < last && save_local[save_i] <= t_abs) {
                sb.reserve(h->n_saved + 1);
                S.interpolate(std::min(save_local[save_i] - t_seg, S.t), sb.row(h->n_saved));
                sb.push_time(save_local[save_i] + shift);
                save_i++;
              }
            } else {
              sb.reserve(h->n_saved + 1);
              S.save_state(sb.row(h->n_saved));
              sb.push_time(t_abs + shift);
            }
            S.select_order();
          }
          // a failed attempt leaves the accept of its last good step (and the copy of that state into the solution
          // buffer, whose time is already recorded) pending: it must not outlive the buffers it points into
          if (failed) { S.flush_accept(); break; }
          // state at the segment end = D[0]
          KIN_HIP(hipMemcpyAsync(S.y.p, S.state_ptr(), N * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        t_seg = seg_end;
        if (ends_at_stop) { apply_rates(h, T_stops, have_table, stop_i); rates_in_force = stop_i; stop_i++; rates_changed = true; }
      }
      if (!failed) {
        // all but the chunk's last save point go to the output, the last only on the final chunk
        // (methods.jl:829-846); its value is the state at the chunk end
        if (chunks && nc == n_chunks - 1 && L > 1 && save_hits_end) save_state_now(save_local[L - 1]);
        next_stop = stop_i;
        break;
      }
      // ---- failure: tighten tolerances and redo this chunk from its start state
      if (progress_every > 0.0 || getenv("KIN_TIMING"))
        fprintf(stderr, "[kin_solve] chunk %lld attempt %d failed with retcode %d at local t = %.6e (h = %.3e, order %d); tolerances %.1e / %.1e\n",
                (long long)nc, attempts, retcode, t_seg + S.t, S.h_abs, S.order, abstol, reltol);
      rates_in_force = -1;
      const double mintol = std::numeric_limits<double>::epsilon();
      if (!p.adaptive_tols || attempts >= 5 || abstol / 10 <= mintol || reltol / 10 <= mintol) break;
      abstol /= 10; reltol /= 10;
      S.set_tols(abstol, reltol);
      S.st.n_retries++;
      S.invalidate_lu_keep_counters();
      S.cache_suspended = true;
      // The retried chunk starts from its start state with NEGATIVE entries set to zero. A chunk can inherit concentrations
      // of -1e-9 (tolerance-sized, 10 x abstol is common) from its predecessor, and mass-action kinetics is unstable under
      // them - a bimolecular term flips its sign - up to a finite-time blow-up of the exact solution from that state: every
      // retry then dies at the same local time whatever its tolerances (3k species, 1 500 K: five attempts, all at
      // t = 7.0e-4). The first attempt of a chunk is untouched (reference semantics); only the rescue path clips.
      launch_clip_negative(N, chunk_start.p, S.y.p, s);
      h->n_saved = saved_at_chunk_start;
      h->sol_t.resize(times_at_chunk_start);
    }
  }
  S.flush_accept();
  h->flush_pending_T(s);     // a temperature no kernel has consumed yet: k is what the caller may read next
  KIN_HIP(hipStreamSynchronize(s));
  S.pre_attempt = nullptr;   // the lambda captures locals of this call
  S.st.final_abstol = abstol;
  S.st.final_reltol = reltol;
  S.st.lu_dense_dim = S.lu.m; S.st.lu_sparse_rows = S.lu.ns; S.st.lu_rounds = S.lu.nrounds;
  S.st.lu_nnz = 2 * S.lu.nnzU + S.lu.ns + (int64_t)S.lu.m * S.lu.m;
  S.st.lu_slots = S.lu_slots;
  S.force_jac_refresh = false;
  S.st.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
  if (getenv("KIN_TIMING"))
    fprintf(stderr, "[kin_solve] wall %.4f s, of which blocked in step syncs %.4f s (the rest is host-side enqueue); "
            "step-end hand-overs that fell back to copy + sync: %lld (fast path %s)\n",
            S.st.wall_seconds, S.sync_wait_s, (long long)S.n_sync_fallbacks, S.fast_sync ? "on" : "off");
  if (getenv("KIN_TIMING"))
    fprintf(stderr, "[kin_solve] corrector iterations per converged attempt: 1:%lld 2:%lld 3:%lld 4:%lld\n", (long long)S.iter_hist[1],
            (long long)S.iter_hist[2], (long long)S.iter_hist[3], (long long)S.iter_hist[4]);
  if (getenv("KIN_TIMING"))
    fprintf(stderr, "[kin_solve] speculatively enqueued steps taken up: %lld of %lld steps, enqueued for nothing: %lld\n",
            (long long)S.n_spec, (long long)S.st.n_steps, (long long)S.n_spec_dead);
  if (stats) *stats = S.st;
  return retcode;
}

// ------------------------------------------------------------------------------------------
// return_integrator=true (methods.jl:175-178, 242-246, 706-709): the initialised integrator is
// handed back instead of being solved; the caller advances it with step!(integ) / solve!(integ)
// and reads integ.t / integ.u. It spans what the reference's integrator spans: the whole tspan for
// solve_chunks=false, the first chunk [0, solve_chunkstep] otherwise; discrete rate updates
// (tstops inside that span) fire when the time reaches them, as the PresetTimeCallback /
// DiscreteCallback of solve_utils.jl:435-509 do.
// ------------------------------------------------------------------------------------------
struct IntegratorState {
  bool active = false, in_segment = false, have_table = false;
  double t_loc0 = 0, t_loc1 = 0, t_seg = 0, seg_end = 0;
  bool ends_at_stop = false;
  std::vector<double> tstops, T_stops;
  std::vector<double> t_nodes, T_nodes;   // continuous rate updates: T(t) = linear interpolation of these (global time)
  int64_t stop_i = 0;
  int retcode = KIN_RETCODE_SUCCESS;
  double T_of(double tg) const {
    const int64_t n = (int64_t)t_nodes.size();
    if (tg <= t_nodes[0]) return T_nodes[0];
    if (tg >= t_nodes[n - 1]) return T_nodes[n - 1];
    const int64_t i = std::upper_bound(t_nodes.begin(), t_nodes.end(), tg) - t_nodes.begin();
    const double dt = t_nodes[i] - t_nodes[i - 1];
    const double th = dt > 0 ? (tg - t_nodes[i - 1]) / dt : 1.0;
    return (1.0 - th) * T_nodes[i - 1] + th * T_nodes[i];
  }
};

void integrator_init(kin_network* h, const kin_params& p, const double* u0, const double* tstops, const double* T_stops,
                     const double* k_table, int64_t n_stops, const double* t_nodes, const double* T_nodes, int64_t n_nodes) {
  const int64_t N = h->host.N, R = h->host.R;
  const bool continuous = n_nodes > 0;
  if (continuous) {
    if (n_stops > 0) throw KinError(ERR_INVALID_ARG, "continuous and discrete rate updates are mutually exclusive");
    if (!t_nodes || !T_nodes || n_nodes < 2) throw KinError(ERR_INVALID_ARG, "need >= 2 (t, T) nodes");
    if (!h->has_arrhenius) throw KinError(ERR_STATE, "continuous rates need the Arrhenius parameters");
    for (int64_t i = 1; i < n_nodes; i++)
      if (!(t_nodes[i] >= t_nodes[i - 1])) throw KinError(ERR_INVALID_ARG, "t_nodes must be non-decreasing");
  }
  if (!(p.tspan0 < p.tspan1)) throw KinError(ERR_INVALID_ARG, "Invalid time span");
  if (!(p.abstol > 0) || !(p.reltol > 0)) throw KinError(ERR_INVALID_ARG, "tolerances must be positive");
  const bool chunks = p.solve_chunks != 0;
  if (chunks) {
    if (!(p.solve_chunkstep > 0)) throw KinError(ERR_INVALID_ARG, "solve_chunkstep must be positive");
    const double q = p.tspan1 / p.solve_chunkstep;
    if (q != std::floor(q) || q < 1) throw KinError(ERR_INVALID_ARG, "Simulation timespan is not divisible by requested chunkwise simulation step size");
  }
  const bool variable = n_stops > 0;
  if (variable) {
    if (!tstops) throw KinError(ERR_INVALID_ARG, "tstops is null");
    if (!k_table && !T_stops) throw KinError(ERR_INVALID_ARG, "need k_table or T_stops with tstops");
    if (!k_table && !h->has_arrhenius) throw KinError(ERR_STATE, "T_stops given but Arrhenius parameters were never set");
    for (int64_t i = 1; i < n_stops; i++)
      if (!(tstops[i] > tstops[i - 1])) throw KinError(ERR_INVALID_ARG, "tstops must be strictly increasing");
  } else if (!continuous && !h->has_rates) {
    throw KinError(ERR_STATE, "rates were never set");
  }
  if (!h->solver) h->solver.reset(new Solver(h));
  if (!h->integ) h->integ.reset(new IntegratorState());
  Solver& S = *h->solver;
  IntegratorState& I = *h->integ;
  hipStream_t s = h->stream;
  S.st = kin_stats{};
  S.invalidate_lu();
  S.accept_pending = false; S.accept_copy = nullptr;
  S.spec = Solver::Spec{};
  S.explicit_mode = false;
  S.ban_negatives = p.ban_negatives != 0;
  S.dtmin = resolve_dtmin(p);
  S.set_tols(p.abstol, p.reltol);
  S.pre_attempt = nullptr;
  S.iters_left = p.maxiters;
  I = IntegratorState{};
  I.have_table = variable && k_table != nullptr;
  if (I.have_table) { h->table.upload(k_table, (size_t)n_stops * R, s); h->table_rows = n_stops; }
  if (variable) {
    I.tstops.assign(tstops, tstops + n_stops);
    if (T_stops) I.T_stops.assign(T_stops, T_stops + n_stops);
  }
  if (continuous) {
    I.t_nodes.assign(t_nodes, t_nodes + n_nodes);
    I.T_nodes.assign(T_nodes, T_nodes + n_nodes);
    h->has_rates = true;
  }
  I.t_loc0 = chunks ? 0.0 : p.tspan0;
  I.t_loc1 = chunks ? p.solve_chunkstep : p.tspan1;
  I.t_seg = I.t_loc0;
  S.y.upload(u0, N, s);
  // rates in force at the start: the last stop at or before it, else the initial conditions (stop 0)
  const double t_start_global = chunks ? 0.0 : p.tspan0;
  while (variable && I.stop_i < n_stops && I.tstops[I.stop_i] <= t_start_global) I.stop_i++;
  if (variable) apply_rates(h, I.T_stops.data(), I.have_table, I.stop_i > 0 ? I.stop_i - 1 : 0);
  I.active = true;
  S.t = 0.0;
  KIN_HIP(hipStreamSynchronize(s));
}

// up to max_steps accepted steps (max_steps <= 0: to the end of the span); returns the number taken
int64_t integrator_step(kin_network* h, int64_t max_steps) {
  if (!h->integ || !h->integ->active) throw KinError(ERR_STATE, "no integrator: call kin_integrator_init first");
  Solver& S = *h->solver;
  IntegratorState& I = *h->integ;
  hipStream_t s = h->stream;
  const int64_t N = h->host.N;
  int64_t taken = 0;
  // continuous rate updates: the Arrhenius rates are re-evaluated at T(global time) of every step attempt, as in
  // solve_entry (the integrator's span starts at global time t_loc0, its segments run in local time)
  if (!I.t_nodes.empty())
    S.pre_attempt = [h, &I](double tau) { h->set_pending_T(I.T_of(I.t_seg + tau)); };
  struct ClearHook {
    kin_network* h; Solver& S;
    ~ClearHook() {
      S.pre_attempt = nullptr; S.hold_speculation = false;
      if (h->k_pending) { try { h->flush_pending_T(h->stream); } catch (...) { h->k_pending = false; } }
    }
  } clear_hook{h, S};
  while (I.retcode == KIN_RETCODE_SUCCESS && I.t_seg < I.t_loc1 && (max_steps <= 0 || taken < max_steps)) {
    if (!I.in_segment) {
      I.seg_end = I.t_loc1;
      I.ends_at_stop = false;
      if (I.stop_i < (int64_t)I.tstops.size() && I.tstops[I.stop_i] < I.t_loc1) { I.seg_end = I.tstops[I.stop_i]; I.ends_at_stop = true; }
      if (S.pre_attempt) S.pre_attempt(0.0);   // rates at the segment start for f0 / J of the restart
      if (!S.restart(0.0, I.seg_end - I.t_seg)) { I.retcode = KIN_RETCODE_UNSTABLE; break; }
      I.in_segment = true;
    }
    const double seg_len = I.seg_end - I.t_seg;
    // the last step of this call gets no speculative batch behind it: other entry points on the handle (kin_set_rates,
    // kin_rates_at, kin_newton_solve) may run before the next call and must not find a half-run step
    S.hold_speculation = max_steps > 0 && taken + 1 >= max_steps;
    StepStatus ss = S.step(seg_len);
    if (S.iters_left < 0) { I.retcode = KIN_RETCODE_MAXITERS; S.flush_accept(); break; }
    if (ss == STEP_DT_MIN) { I.retcode = KIN_RETCODE_DTLESSTHANMIN; S.flush_accept(); break; }
    if (ss == STEP_UNSTABLE) { I.retcode = KIN_RETCODE_UNSTABLE; S.flush_accept(); break; }
    S.select_order();
    taken++;
    if (S.t >= seg_len) {   // segment finished: state = D[0]; switch the rates at a tstop
      KIN_HIP(hipMemcpyAsync(S.y.p, S.state_ptr(), N * sizeof(double), hipMemcpyDeviceToDevice, s));
      I.t_seg = I.seg_end;
      I.in_segment = false;
      S.t = 0.0;
      if (I.ends_at_stop) { apply_rates(h, I.T_stops.data(), I.have_table, I.stop_i); I.stop_i++; }
    }
  }
  h->flush_pending_T(s);
  KIN_HIP(hipStreamSynchronize(s));
  return taken;
}

void integrator_state(kin_network* h, double* t, double* u, int32_t* retcode, kin_stats* stats) {
  if (!h->integ || !h->integ->active) throw KinError(ERR_STATE, "no integrator: call kin_integrator_init first");
  Solver& S = *h->solver;
  IntegratorState& I = *h->integ;
  if (t) *t = I.in_segment ? I.t_seg + S.t : I.t_seg;
  if (u) {
    if (I.in_segment) S.flush_accept();
    (I.in_segment ? S.D : S.y).download(u, h->host.N, h->stream);   // D[0] = state after the last accepted step
    KIN_HIP(hipStreamSynchronize(h->stream));
  }
  if (retcode) *retcode = I.retcode;
  if (stats) { *stats = S.st; stats->final_abstol = S.atol; stats->final_reltol = S.rtol; }
}

void solution_max(kin_network* h, double* out_umax) {
  if (h->n_saved <= 0) throw KinError(ERR_STATE, "no solution stored");
  // (the stored solution may come from the resident integrator: no Solver object then; the scratch vector is the handle's)
  h->du.alloc(h->host.N);
  launch_colmax((int)h->host.N, h->n_saved, h->d_sol_u.p, h->du.p, h->stream);
  h->du.download(out_umax, h->host.N, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
}

// scatter b into the permuted solve vector / gather x back (diagnostic path only)
void newton_solve(kin_network* h, double c, const double* u, const double* b, double* x) {
  if (!h->solver) h->solver.reset(new Solver(h));
  Solver& S = *h->solver;
  hipStream_t s = h->stream;
  const int N = S.N;
  S.flush_accept();
  S.spec = Solver::Spec{};
  S.y.upload(u, N, s);
  S.eval_jac(S.y.p);
  S.lu.factor(c, S.jv.p, 0, &S.ctrl.p->lu_bad, s);
  S.cur_slot = 0;
  std::vector<int32_t> yl(N), xl(N);
  S.lu.yloc.download(yl.data(), N, s);
  S.lu.xloc.download(xl.data(), N, s);
  std::vector<double> W(S.lu.w_size);
  KIN_HIP(hipStreamSynchronize(s));
  std::vector<double> stage(N);
  // the solve vectors live at the tail of W: write b there element by element
  std::vector<double> tail((size_t)(S.lu.off_x + S.lu.mpad + 8 - S.lu.off_y), 0.0);   // the solve vectors only
  for (int i = 0; i < N; i++) tail[yl[i] - S.lu.off_y] = b[i];
  KIN_HIP(hipMemcpyAsync(S.lu.slots[0].W.p + S.lu.off_y, tail.data(), tail.size() * sizeof(double), hipMemcpyHostToDevice, s));
  S.lu.solve(nullptr, 0, s);
  KIN_HIP(hipMemcpyAsync(tail.data(), S.lu.slots[0].W.p + S.lu.off_y, tail.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  int bad = 0;
  KIN_HIP(hipMemcpyAsync(&bad, &S.ctrl.p->lu_bad, sizeof(int), hipMemcpyDeviceToHost, s));
  KIN_HIP(hipStreamSynchronize(s));
  S.invalidate_lu();
  if (bad) {
    KIN_HIP(hipMemsetAsync(&S.ctrl.p->lu_bad, 0, sizeof(int), s));
    KIN_HIP(hipStreamSynchronize(s));
    throw KinError(ERR_SOLVE_FAILED, "vanishing pivot in the factorisation of I - c J (static diagonal pivoting)");
  }
  for (int i = 0; i < N; i++) x[i] = tail[xl[i] - S.lu.off_y];
}

}  // namespace kin

kin_network::kin_network() {}
kin_network::~kin_network() {
  for (kin_network* r : replicas) delete r;
  replicas.clear();
  integ.reset();
  solver.reset();
  if (stream) (void)hipStreamDestroy(stream);
}
