// The resident integrator's controller: the whole of kin_solve's orchestration (chunk loop, discrete rate updates, save
// grid, adaptive_solve! retries - reference src/solving/methods.jl:185-303, 717-865; solve_utils.jl:376-424, 435-509) and of
// the BDF integrator that stands in for the reference's CVODE_BDF (solver.cpp: Solver::restart / step / select_order /
// interpolate, including the LU cache and its guards), written ONCE against a backend `B` that supplies the vector
// operations, the Jacobian, the factorisation and the corrector iteration:
//   * resident.hip compiles it for the device: ONE 512-thread workgroup owns one trajectory for the whole solve - every
//     thread runs this controller redundantly on identical scalars (reduction results are broadcast), the backend's
//     operations are workgroup-wide phases separated by barriers; no host round trip, no kernel boundary per step;
//   * tests/native/resident_host.cpp compiles it for the CPU with a sequential backend over the same tables: the
//     controller's logic is checked without a GPU (test infrastructure, not a product path).
// The host-driven integrator of solver.cpp stays the path for large networks; both implement the same algorithm with the
// same constants. At the default tolerances (1e-10 / 1e-8) their trajectories agree within the step-sequence
// tolerance (<= 201 units over the 312 solves of profiles/r05_robustness_resident.jsonl), not bit for bit. At rtol = 1e-10 every
// implementation runs on the rounding floor of the right-hand side and they differ by the accuracy of their LINEAR ALGEBRA: against
// a Radau truth (tests/golden/truth_tight_200.npz) the CPU port's pivoted LU lands at rms 61 / max 378 tight units in 4 073
// steps, this controller over the in-workgroup factorisation at 55 / 757 in 5 656, the host-driven path's explicit inverses at
// 394 / 5 572 in 9 785 with 6 172 corrector failures (profiles/r05_tight_tol_truth.jsonl; tests/test_gpu_resident.py bounds both).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define KIN_HD __host__ __device__
#else
#define KIN_HD
#endif

namespace kin {

constexpr int RES_MAX_ORDER = 5;
constexpr int RES_NEWTON_MAXITER = 4;
constexpr int RES_D_ROWS = RES_MAX_ORDER + 3;
constexpr int RES_MAX_SLOTS = 64;   // one LU-cache slot per lane of a wavefront (resident.hip keeps the slot table in registers)

enum : int { RES_RET_SUCCESS = 0, RES_RET_MAXITERS = 1, RES_RET_DTLESSTHANMIN = 2, RES_RET_UNSTABLE = 3 };   // = KIN_RETCODE_*

// corrector tolerance as a fraction of the error weight: solver_kernels.hpp bdf_newton_frac has the rule and its measurements
KIN_HD inline double res_newton_frac(double rtol) { return fmin(0.1, fmax(0.03, 1e-10 / rtol)); }

// what a solve needs besides the network (plain data; pointers are device pointers in the product, host pointers in the test)
// largest power of ten <= h by exact IEEE operations only (solver.cpp: decade_floor)
KIN_HD inline double res_decade_floor(double h) {
  if (!(h > 0.0) || !(h < 1e300)) return h;
  double p = 1.0;
  while (p > h) p /= 10.0;
  while (p * 10.0 <= h) p *= 10.0;
  return p;
}

struct ResParams {
  double tspan0, tspan1, abstol, reltol, chunkstep, dtmin;
  int32_t solve_chunks, adaptive_tols, ban_negatives, save_hits_end;
  int64_t maxiters, n_chunks;
  int32_t L;                       // points of the per-chunk (or whole-span) save grid
  const double* save_local;        // L local save times (solver.cpp: save_local)
  int32_t n_stops, rate_mode;      // rate_mode 0: static k; 1: k_table[n_stops][R]; 2: Arrhenius at T_stops[n_stops]
  const double* tstops;
  // integrator settings (the same defaults and environment switches as solver.cpp)
  double lu_band, reuse_rate_max, crate_dy_max, lu_drift_max, newton_frac;
  int64_t crate_max_age, lu_max_age;
  int32_t n_slots, carry_rate;
  int64_t sol_cap;                 // rows of the solution buffer
  int32_t profile, pad0;           // device: fill ResResult::prof (KIN_RESIDENT_PROFILE)
};

struct ResStats {
  int64_t n_steps, n_rejected, n_rhs, n_jac, n_factor, n_linsolve, n_newton_fail, n_chunks, n_restarts, n_retries,
      n_lu_reused, n_bad_pivot, n_lu_dropped;
};

struct ResResult {
  int32_t retcode, pad;
  int64_t n_saved;
  double final_abstol, final_reltol;
  ResStats st;
  int64_t prof[20];   // device only: 10 ns ticks per phase kind (resident.hip: ProfId), 0 in the CPU replay
};

// rms(y / w), rms(f0 / w), rms((f1 - f0) / w), max |f0| / (0.1 |y| + w) over the species, w = atol + rtol |y|
struct ResNorms { double d0, d1, d2, dmax; int nonfinite; };
struct ResSums { double s, se, sm, sp, neg; };   // update, error test of order / order - 1 / order + 1, negative entries

KIN_HD inline double res_inf() { return HUGE_VAL; }
// spacing of the doubles above x (x >= 0, finite): what std::nextafter(x, inf) - x gives on the host
KIN_HD inline double res_ulp_above(double x) { return nextafter(x, res_inf()) - x; }

// one corrector attempt: what the controller hands to the backend, and what it gets back
struct ResCorrIn {
  int32_t slot, order;
  double c, upd, rate_max, crate0, tol_first, newton_tol, dy_first_max, ec, ec_m, ec_p, atol, rtol, alpha_o;
};
struct ResAttempt { bool done, converged, nonfinite, any_negative, deep_negative; int n_iter; double err, err_m, err_p, crate; };
// solver_kernels.hpp BDF_NEG_DEEP / BDF_NEG_MARK: an accepted step with a species below -1e3 error weights ends the segment
constexpr double RES_NEG_DEEP = 1e3, RES_NEG_MARK = 4294967296.0;

// Predictor + corrector iterations until decided: the decisions of newton_decide (solver_kernels.hip), taken in sequence.
// `I` supplies predict_inner / newton_iter_inner (the CPU replay: the backend itself; the device: the phase's own inlined
// operations, run by all wavefronts).
template <class I>
KIN_HD ResAttempt res_corrector_loop(I& b, const ResCorrIn& in, const double* gamma, int n_species) {
  ResAttempt a{false, false, false, false, false, 0, 0.0, 0.0, 0.0, 1.0};
  b.predict_inner(in.order, gamma, in.alpha_o, in.atol, in.rtol);
  const double N = (double)n_species;
  double crate = in.crate0, dy_old = 0.0;
  for (int it = 0; it < RES_NEWTON_MAXITER && !a.done; it++) {
    const ResSums q = b.newton_iter_inner(in.slot, in.c, in.upd, in.order, in.ec, in.ec_m, in.ec_p, in.atol, in.rtol);
    const double dy_norm = sqrt(q.s / N);
    const bool nonfinite = !(fabs(q.s) <= 1.79769313486231570815e308);   // !isfinite
    const bool have_rate = it > 0;
    const double rate = have_rate ? dy_norm / dy_old : 0.0;
    if (have_rate && !nonfinite) crate = (0.3 * crate > rate) ? 0.3 * crate : rate;
    bool diverged = nonfinite;
    if (!diverged && have_rate) {
      double rp = rate;
      for (int e = 1; e < RES_NEWTON_MAXITER - it; e++) rp *= rate;
      if (rate >= in.rate_max || rp / (1.0 - rate) * dy_norm > in.newton_tol) diverged = true;
    }
    a.n_iter = it + 1;
    a.done = true;
    if (diverged) { a.nonfinite = nonfinite; }
    else if (dy_norm == 0.0 || (have_rate && rate / (1.0 - rate) * dy_norm < in.newton_tol) ||
             (!have_rate && (dy_norm < in.newton_tol || (in.crate0 < 1.0 && dy_norm <= in.dy_first_max &&
                                                         in.crate0 / (1.0 - in.crate0) * dy_norm < in.tol_first)))) {
      a.converged = true;
    } else {
      dy_old = dy_norm;
      a.done = it == RES_NEWTON_MAXITER - 1;
    }
    if (a.converged) {
      a.err = sqrt(q.se / N); a.err_m = sqrt(q.sm / N); a.err_p = sqrt(q.sp / N);
      a.any_negative = q.neg > 0.0;
      a.deep_negative = q.neg >= RES_NEG_MARK;
      if (!(fabs(q.se) <= 1.79769313486231570815e308)) a.nonfinite = true;
    }
  }
  a.crate = crate;
  return a;
}

template <class B>
struct ResidentBdf {
  B& b;
  const ResParams& P;
  double gamma[RES_MAX_ORDER + 1], alpha[RES_MAX_ORDER + 1], errc[RES_MAX_ORDER + 2];
  // integrator state (Solver, solver.cpp)
  double t = 0, h_abs = 0, atol = 0, rtol = 0, newton_tol = 0, dtmin = 0, fail_score = 0;
  int order = 1, n_equal = 0, cur_slot = 0;
  bool lu_valid = false, jac_current = false, force_jac_refresh = false, force_fresh_lu = false, cache_suspended = false,
       slot_is_fresh = false, pending_order_change = false,
       first_selection = false;   // the next step-size selection is the first since a (re)initialisation: growth cap 1e4 (CVODE's ETAMX1), 10 afterwards
  int64_t steps_since_jac = 0, jac_stamp_now = 0, use_clock = 0, iters_left = 0;
  double err_m = 0, err_p = 0, err_o = 0, safety_o = 0.9;
  ResStats st;

  KIN_HD ResidentBdf(B& bb, const ResParams& pp) : b(bb), P(pp) {
    const double KAPPA[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    gamma[0] = 0.0;
    for (int j = 1; j <= RES_MAX_ORDER; j++) gamma[j] = gamma[j - 1] + 1.0 / j;
    for (int j = 0; j <= RES_MAX_ORDER; j++) alpha[j] = (1.0 - KAPPA[j]) * gamma[j];
    for (int j = 0; j <= RES_MAX_ORDER; j++) errc[j] = KAPPA[j] * gamma[j] + 1.0 / (j + 1);
    errc[RES_MAX_ORDER + 1] = 0.0;
    st = ResStats{};
  }

  KIN_HD void set_tols(double a, double r) {
    atol = a; rtol = r;
    const double lo = 10.0 * 2.220446049250313e-16 / r;
    const double frac = P.newton_frac > 0.0 ? P.newton_frac : res_newton_frac(r);
    newton_tol = lo > frac ? lo : frac;
  }

  // work matrices of change_D: MEMBERS, not locals - on the device the controller object lives in LDS, dynamically indexed
  // local arrays would live in scratch (= global memory: a dependent store / load pair there costs microseconds)
  double wM[6][6], wR[6][6], wU[6][6], wRU[6][6], wp[RES_MAX_ORDER + 1];
  KIN_HD void compute_R(int ord, double factor, double R[6][6]) {
    double (&M)[6][6] = wM;
    for (int i = 0; i <= ord; i++)
      for (int j = 0; j <= ord; j++) M[i][j] = 0.0;
    for (int j = 0; j <= ord; j++) M[0][j] = 1.0;
    for (int i = 1; i <= ord; i++)
      for (int j = 1; j <= ord; j++) M[i][j] = ((double)i - 1.0 - factor * (double)j) / (double)i;
    for (int j = 0; j <= ord; j++) {
      double p = 1.0;
      for (int i = 0; i <= ord; i++) { p *= M[i][j]; R[i][j] = p; }
    }
  }
  KIN_HD void change_D(int ord, double factor) {
    double (&R)[6][6] = wR; double (&U)[6][6] = wU; double (&RU)[6][6] = wRU;
    compute_R(ord, factor, R);
    compute_R(ord, 1.0, U);
    for (int a = 0; a <= ord; a++)
      for (int q2 = 0; q2 <= ord; q2++) {
        double v = 0.0;
        for (int q = 0; q <= ord; q++) v += R[a][q] * U[q][q2];
        RU[a][q2] = v;
      }
    b.change_D(ord, RU);
  }

  KIN_HD void eval_jac() { b.eval_jac_y(); st.n_jac++; lu_valid = false; steps_since_jac = 0; jac_stamp_now = st.n_restarts; }
  KIN_HD void predict() { b.predict(order, gamma, alpha[order], atol, rtol); }

  // Warm continuation at a chunk start whose rates did not change (kin_params.solve_chunks == 2; Solver::resume): the system
  // is autonomous and chunks run in local time, so difference history, order and step size stay valid. The Jacobian is
  // evaluated at the chunk's first state and the LU cache's drift guard runs, as at a re-initialisation.
  KIN_HD void resume() {
    t = 0.0;
    st.n_restarts++;
    eval_jac();
    jac_current = true;
    if (P.lu_band > 0.0 && P.lu_drift_max > 0.0) st.n_lu_dropped += b.drift_check(P.lu_drift_max);
  }

  // (re)start at segment-local time 0 from the state in y: order 1, fresh initial step, fresh Jacobian (Solver::restart).
  // Initial step = CVODE's (cvode.c: cvHin, cvUpperBoundH0, cvYddNorm - the documented solver of the reference,
  // docs/src/getting-started.md:69, re-initialised at every chunk start and rate update, methods.jl:260, 819): the h with
  // ||h^2 y'' / 2||_WRMS = 1, y'' from a difference quotient of f along the Euler direction, iterated (at most 4 evaluations)
  // until two successive estimates agree within a factor of 2, halved (H_BIAS), kept inside [hlb, hub]: hlb = 100 ulp of the
  // segment, hub = a tenth of the segment but no step over which ANY component moves by more than a tenth of itself plus its
  // error weight (`dmax` of the norms). Then rounded DOWN to a power of ten: the step size climbs through the same values after
  // every restart, so the iteration matrices of the previous segment's climb are found in the LU cache again.
  KIN_HD bool restart(double t_bound) {
    t = 0.0;
    st.n_restarts++;
    eval_jac();
    jac_current = true;
    if (P.lu_band > 0.0 && P.lu_drift_max > 0.0) st.n_lu_dropped += b.drift_check(P.lu_drift_max);
    b.rhs_y_to_f0(); st.n_rhs++;
    ResNorms n0 = b.norms(false, atol, rtol);
    if (n0.nonfinite) return false;
    const double tdist = fabs(t_bound);
    const double hlb = 100.0 * 2.220446049250313e-16 * tdist;
    double hub = 0.1 * tdist;
    if (hub * n0.dmax > 1.0) hub = 1.0 / n0.dmax;
    double hg = sqrt(hlb * hub), hnew = hg;
    if (hub >= hlb) {
      for (int count = 1; count <= 4; count++) {
        b.ytmp_axpy(hg);
        b.rhs_ytmp_to_f1(); st.n_rhs++;
        ResNorms n1 = b.norms(true, atol, rtol);
        if (n1.nonfinite) return false;
        const double ydd = n1.d2 / hg;
        hnew = ydd * hub * hub > 2.0 ? sqrt(2.0 / ydd) : sqrt(hg * hub);
        if (count == 4) break;
        const double hrat = hnew / hg;
        if (hrat > 0.5 && hrat < 2.0) break;
        if (count > 1 && hrat > 2.0) { hnew = hg; break; }
        hg = hnew;
      }
    }
    double h0 = 0.5 * hnew;
    h0 = h0 < hlb ? hlb : h0;
    h0 = h0 > hub ? hub : h0;
    h0 = h0 < tdist ? h0 : tdist;
    h_abs = res_decade_floor(h0);
    b.init_D(false, h_abs);
    order = 1; n_equal = 0; fail_score = 0.0;
    first_selection = true;
    return true;
  }

  KIN_HD void reset_history() {
    b.ytmp_from_D0();
    b.rhs_ytmp_to_f0(); st.n_rhs++;
    b.init_D(true, h_abs);
    order = 1; n_equal = 0; lu_valid = false; fail_score = 0.0;
    b.slots_invalidate(false);
    force_jac_refresh = true;
  }

  KIN_HD void invalidate_lu_keep_counters() { b.slots_invalidate(false); lu_valid = false; cur_slot = 0; force_fresh_lu = false; }
  KIN_HD void invalidate_lu() {
    b.slots_invalidate(true);
    lu_valid = false; cur_slot = 0; use_clock = 0; force_fresh_lu = false; steps_since_jac = 0; cache_suspended = false;
  }

  KIN_HD bool crate_fresh(int slot) const {
    return P.carry_rate && b.slot_crate(slot) < 1.0 && b.slot_crate_restart(slot) == st.n_restarts &&
           st.n_steps - b.slot_crate_step(slot) <= P.crate_max_age;
  }

  // returns true when a pivot vanished
  KIN_HD bool factor_into(int slot, double c) {
    const bool bad = b.factor(slot, c, P.lu_band > 0.0);
    b.slot_made(slot, c, ++use_clock, jac_stamp_now, st.n_steps - steps_since_jac);
    cur_slot = slot;
    st.n_factor++;
    return bad;
  }

  // predictor + corrector iterations until decided (one backend operation: on the device one phase, no round trip through
  // the controller between the iterations)
  KIN_HD ResAttempt corrector(double c) {
    ResCorrIn in;
    const double cf = b.slot_c_fact(cur_slot);
    in.slot = cur_slot; in.order = order; in.c = c;
    in.upd = cf != c ? 2.0 / (1.0 + c / cf) : 1.0;
    in.rate_max = (P.lu_band > 0.0 && !cache_suspended && !slot_is_fresh) ? P.reuse_rate_max : 1.0;
    in.crate0 = P.carry_rate ? b.slot_crate(cur_slot) : 1.0;
    in.tol_first = crate_fresh(cur_slot) ? newton_tol : -1.0;
    in.newton_tol = newton_tol; in.dy_first_max = P.crate_dy_max;
    in.ec = errc[order]; in.ec_m = order > 1 ? errc[order - 1] : 0.0; in.ec_p = errc[order + 1];
    in.atol = atol; in.rtol = rtol; in.alpha_o = alpha[order];
    return b.corrector(in, gamma);
  }

  enum StepStatus { STEP_OK = 0, STEP_DT_MIN = 1, STEP_UNSTABLE = 2 };

  // one accepted step towards t_bound (Solver::step without the asynchronous machinery)
  KIN_HD StepStatus step(double t_bound) {
    bool accepted = false, first_attempt = true;
    double safety = 0.9, err_norm = 0.0, t_new = t;
    ResAttempt a{};
    while (!accepted) {
      if (iters_left-- <= 0) return STEP_OK;   // caller checks iters_left < 0 -> MaxIters
      const double ulp10 = 10.0 * res_ulp_above(t);
      const double min_step = dtmin > ulp10 ? dtmin : ulp10;
      if (h_abs < min_step) {
        if (!first_attempt) return STEP_DT_MIN;
        change_D(order, min_step / h_abs);
        h_abs = min_step; n_equal = 0; lu_valid = false;
      }
      first_attempt = false;
      t_new = t + h_abs;
      if (t_new - t_bound > 0.0) {
        t_new = t_bound;
        change_D(order, fabs(t_new - t) / h_abs);
        n_equal = 0; lu_valid = false;
      }
      const double hh = t_new - t;
      h_abs = fabs(hh);
      const double c = hh / alpha[order];
      bool converged = false;
      if (force_jac_refresh) { predict(); eval_jac(); jac_current = true; force_jac_refresh = false; }
      bool fresh = false, bad = false;
      const double band = cache_suspended ? 0.0 : P.lu_band;
      if (band > 0.0) {
        const int hit = b.nearest_slot(c, band, st.n_restarts, P.lu_max_age);
        if (hit >= 0 && !force_fresh_lu) { cur_slot = hit; b.slot_touch(hit, ++use_clock); st.n_lu_reused++; }
        else {
          if (force_fresh_lu && !jac_current && steps_since_jac > 20) { predict(); eval_jac(); jac_current = true; }
          bad = factor_into(hit >= 0 ? hit : b.victim_slot(st.n_restarts, P.lu_max_age, P.n_slots), c);
          fresh = jac_current;
        }
        force_fresh_lu = false;
      } else if (!lu_valid) {
        bad = factor_into(0, c);
        lu_valid = true;
        fresh = jac_current;
      }
      for (;;) {
        slot_is_fresh = fresh || b.slot_c_fact(cur_slot) == c;
        a = corrector(c);
        st.n_rhs += a.n_iter; st.n_linsolve += a.n_iter;
        if (a.n_iter > 1) b.slot_rate(cur_slot, a.crate, st.n_steps, st.n_restarts);
        converged = a.done && a.converged && !a.nonfinite;
        if (bad) {
          // a pivot of the factorisation in hand vanished (static pivoting): the slot is dropped, the step is retried at
          // half the size from a Jacobian at its own predictor
          b.slot_drop(cur_slot);
          lu_valid = false; force_jac_refresh = true; st.n_bad_pivot++;
          converged = false;
          break;
        }
        if (converged) break;
        st.n_newton_fail++;
        if (band > 0.0) {
          if (fresh) break;
          if (!jac_current) { predict(); eval_jac(); jac_current = true; }
          bad = factor_into(cur_slot, c);
          fresh = true;
          continue;
        }
        if (jac_current) break;
        predict(); eval_jac(); jac_current = true;
        bad = factor_into(0, c);
        lu_valid = true;
      }
      if (!converged || (P.ban_negatives && a.any_negative)) {
        // a failed corrector cuts the step to a quarter and does not count towards the history reset (CVODE: ETACF = 0.25,
        // history rebuilt only after repeated error-test failures - solver.cpp: cf_eta / cf_resets); a banned negative state
        // (isoutofdomain, methods.jl:169-171) halves it and counts
        const double eta = !converged ? 0.25 : 0.5;
        h_abs *= eta;
        change_D(order, eta);
        n_equal = 0; lu_valid = false;
        st.n_rejected++;
        first_selection = false;   // (CVODE: any failed attempt sets etamax = 1, the first step's 1e4 is gone)
        if (converged) fail_score += 1.0;
        if (fail_score >= 3.0 && order > 1) reset_history();
        continue;
      }
      if (band > 0.0 && !fresh && a.n_iter >= RES_NEWTON_MAXITER) b.slot_drop(cur_slot);
      safety = 0.9 * (2.0 * RES_NEWTON_MAXITER + 1.0) / (2.0 * RES_NEWTON_MAXITER + a.n_iter);
      err_norm = a.err;
      if (err_norm > 1.0) {
        const double f0 = safety * pow(err_norm, -1.0 / (order + 1));
        const double factor = f0 > 0.2 ? f0 : 0.2;
        h_abs *= factor;
        change_D(order, factor);
        n_equal = 0;
        force_fresh_lu = band > 0.0;
        st.n_rejected++;
        first_selection = false;
        fail_score += 1.0;
        if (fail_score >= 3.0 && order > 1) reset_history();
      } else {
        if (a.deep_negative) return STEP_UNSTABLE;   // solver_kernels.hpp BDF_NEG_DEEP: the negative excursion, given up early
        accepted = true;
      }
    }
    st.n_steps++;
    steps_since_jac++;
    fail_score = fail_score - 0.2 > 0.0 ? fail_score - 0.2 : 0.0;
    n_equal++;
    t = t_new;
    b.accept(order);
    jac_current = false;
    pending_order_change = (n_equal >= order + 1);
    if (pending_order_change) {
      err_m = order > 1 ? a.err_m : res_inf();
      err_p = order < RES_MAX_ORDER ? a.err_p : res_inf();
      err_o = err_norm;
      safety_o = safety;
    }
    return STEP_OK;
  }

  KIN_HD void select_order() {
    if (!pending_order_change) return;
    pending_order_change = false;
    const double norms[3] = {err_m, err_o, err_p};
    double best = -1.0;
    int arg = 1;
    for (int i = 0; i < 3; i++) {
      double f;
      if (norms[i] == 0.0) f = res_inf();
      else if (norms[i] == res_inf()) f = 0.0;
      else f = pow(norms[i], -1.0 / (order + i));
      if (f > best) { best = f; arg = i; }
    }
    order += arg - 1;
    const double f1 = safety_o * best;
    const double cap = first_selection ? 1e4 : 10.0;
    first_selection = false;
    const double factor = f1 < cap ? f1 : cap;
    h_abs *= factor;
    change_D(order, factor);
    n_equal = 0;
    lu_valid = false;
  }

  // dense output of the step that ended at t into solution row `row`
  KIN_HD void interpolate(double ts, int64_t row) {
    double (&p)[RES_MAX_ORDER + 1] = wp;
    double prod = 1.0;
    p[0] = 0.0;
    for (int j = 0; j < order; j++) {
      prod *= (ts - (t - h_abs * j)) / (h_abs * (1.0 + j));
      p[j + 1] = prod;
    }
    b.interp(order, p, row);
  }

  // ---- the driver (solve_entry, solver.cpp); returns the result block
  KIN_HD ResResult run() {
    const bool chunks = P.solve_chunks != 0;
    const bool variable = P.n_stops > 0;
    const int L = P.L;
    double abstol = P.abstol, reltol = P.reltol;
    set_tols(abstol, reltol);
    dtmin = P.dtmin;
    invalidate_lu();
    b.load_u0();
    int64_t next_stop = 0, n_saved = 0, rates_in_force = -1;
    bool have_history = false, rates_changed = false;   // warm continuation across chunk starts (solve_chunks == 2)
    int retcode = RES_RET_SUCCESS;
    for (int64_t nc = 0; nc < P.n_chunks && retcode == RES_RET_SUCCESS; nc++) {
      st.n_chunks++;
      cache_suspended = false;
      const double t_start_global = chunks ? P.chunkstep * (double)nc : P.tspan0;
      const double t_end_global = chunks ? t_start_global + P.chunkstep : P.tspan1;
      const double shift = chunks ? (double)nc * P.chunkstep : 0.0;
      const double t_loc0 = chunks ? 0.0 : P.tspan0;
      const double t_loc1 = chunks ? P.chunkstep : P.tspan1;
      const int64_t stop_first = next_stop;
      b.chunk_start_from_y();
      const int64_t saved_at_chunk_start = n_saved;
      int attempts = 0;
      for (;;) {   // adaptive_solve! (solve_utils.jl:376-424)
        attempts++;
        if (attempts > 1) have_history = false;   // a retry starts cold from the chunk's first state
        retcode = RES_RET_SUCCESS;
        iters_left = P.maxiters;
        int64_t stop_i = stop_first;
        while (variable && stop_i < P.n_stops && P.tstops[stop_i] <= t_start_global) stop_i++;
        if (variable) {
          const int64_t want = stop_i > 0 ? stop_i - 1 : 0;
          if (want != rates_in_force) { b.apply_rates(want); rates_in_force = want; rates_changed = true; }
        }
        int save_i = 0;
        double t_seg = t_loc0;
        bool failed = false;
        if (n_saved < P.sol_cap) { b.save_y(n_saved, P.save_local[0] + shift); }
        n_saved++;
        save_i = 1;
        while (t_seg < t_loc1 && !failed) {
          double seg_end = t_loc1;
          bool ends_at_stop = false;
          if (variable && stop_i < P.n_stops && P.tstops[stop_i] < t_end_global) {
            const double loc = P.tstops[stop_i] - shift;
            if (loc < t_loc1) { seg_end = loc; ends_at_stop = true; }
          }
          if (seg_end > t_seg) {
            const double seg_len = seg_end - t_seg;
            if (P.solve_chunks == 2 && have_history && !rates_changed) resume();
            else if (!restart(seg_len)) { retcode = RES_RET_UNSTABLE; failed = true; break; }
            have_history = true;
            rates_changed = false;
            while (t < seg_len) {
              const StepStatus ss = step(seg_len);
              if (iters_left < 0) { retcode = RES_RET_MAXITERS; failed = true; break; }
              if (ss == STEP_DT_MIN) { retcode = RES_RET_DTLESSTHANMIN; failed = true; break; }
              if (ss == STEP_UNSTABLE) { retcode = RES_RET_UNSTABLE; failed = true; break; }
              const double t_abs = t >= seg_len ? seg_end : t_seg + t;
              const int last = (chunks && !(nc == P.n_chunks - 1 && !P.save_hits_end)) ? L - 1 : L;
              while (save_i < last && P.save_local[save_i] <= t_abs) {
                const double tl = P.save_local[save_i] - t_seg;
                if (n_saved < P.sol_cap) { interpolate(tl < t ? tl : t, n_saved); b.set_time(n_saved, P.save_local[save_i] + shift); }
                n_saved++;
                save_i++;
              }
              select_order();
            }
            if (failed) break;
            b.y_from_D0();
          }
          t_seg = seg_end;
          if (ends_at_stop) { b.apply_rates(stop_i); rates_in_force = stop_i; stop_i++; rates_changed = true; }
        }
        if (!failed) {
          if (chunks && nc == P.n_chunks - 1 && L > 1 && P.save_hits_end) {
            if (n_saved < P.sol_cap) b.save_y(n_saved, P.save_local[L - 1] + shift);
            n_saved++;
          }
          next_stop = stop_i;
          break;
        }
        // failure: tighten the tolerances and redo this chunk from its (clipped) start state
        rates_in_force = -1;
        const double mintol = 2.220446049250313e-16;
        if (!P.adaptive_tols || attempts >= 5 || abstol / 10 <= mintol || reltol / 10 <= mintol) break;
        abstol /= 10; reltol /= 10;
        set_tols(abstol, reltol);
        st.n_retries++;
        invalidate_lu_keep_counters();
        cache_suspended = true;
        b.y_from_chunk_start_clipped();
        n_saved = saved_at_chunk_start;
      }
    }
    ResResult r;
    r.retcode = retcode; r.pad = 0; r.n_saved = n_saved; r.final_abstol = abstol; r.final_reltol = reltol; r.st = st;
    for (int i = 0; i < 20; i++) r.prof[i] = 0;
    b.profile_out(r.prof);
    return r;
  }
};

}  // namespace kin
