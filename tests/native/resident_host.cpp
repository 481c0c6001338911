// TEST INFRASTRUCTURE - not a product path, never loaded by kinetica_jl_amd.
// CPU replay of the resident integrator: kinetica_jl_amd/csrc/resident_core.hpp (the controller the GPU kernel runs)
// compiled with a sequential backend over the same host-side tables the device gets (network tables, gather plans, the
// symbolic factorisation of SparseLU::analyze kept on the host). Checks, without a GPU, (1) the controller's logic against
// oracle/cpu_bdf.cpp (the same algorithm written independently) and the committed truths, and (2) the symbolic LU analysis
// of lu.cpp numerically (factor + solve against a dense solve), which no other CPU test reaches.
// Built by tests/native/Makefile into libkin_resident_host.so; links libkinetica_hip.so for the host-side C++ it shares.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../kinetica_jl_amd/csrc/lu.hpp"
#include "../../kinetica_jl_amd/csrc/network.hpp"
#include "../../kinetica_jl_amd/csrc/resident_core.hpp"
#include "../../kinetica_jl_amd/csrc/resident_setup.hpp"

using namespace kin;

namespace {

struct Ex { const double* psi = nullptr; const double* d = nullptr; double cscal = 0.0; };

// the arithmetic of segsum_kernel / seg_run, row by row (summation order inside a row is not replicated: rounding-level)
void seg_run_host(int op, const SegPlanHost& p, const double* src, double* out, const Ex& ex) {
  const bool prod = op == SEG_PROD_SUB || op == SEG_PROD_SUB_DIV || op == SEG_PROD_AUXSUB || op == SEG_PROD_SET || op == SEG_PROD_NEG;
  const bool impl = p.val_base >= 0;
  auto store = [&](int32_t dst, int32_t aux, double acc) {
    switch (op) {
      case SEG_COEF_SET: out[dst] = acc; break;
      case SEG_PROD_SUB: out[dst] = out[dst] - acc; break;
      case SEG_PROD_SUB_DIV: out[dst] = (out[dst] - acc) / src[aux]; break;
      case SEG_PROD_AUXSUB: out[dst] = src[aux] - acc; break;
      case SEG_PROD_SET: out[dst] = acc; break;
      case SEG_PROD_NEG: out[dst] = -acc; break;
      default: out[dst] = ex.cscal * acc - ex.psi[aux] - ex.d[aux];
    }
  };
  auto term = [&](bool ell, int32_t e) -> double {
    if (prod) {
      const int32_t ib = ell ? p.ell_b[e] : p.long_b[e];
      if (ib < 0) return 0.0;
      const int32_t ia = impl ? (ell ? p.val_base + e : p.val_base + p.ell_total + e) : (ell ? p.ell_a[e] : p.long_a[e]);
      return src[ia] * src[ib];
    }
    const float c = ell ? p.ell_c[e] : p.long_c[e];
    if (c == 0.0f) return 0.0;
    return (double)c * src[ell ? p.ell_a[e] : p.long_a[e]];
  };
  // operands that do not depend on the sums are read before any store of the same plan (as seg_pre does per task); rows of
  // one plan never read what another row of the same plan writes, except in place (out[dst] itself)
  for (int g = 0; g < p.n_groups(); g++)
    for (int lane = 0; lane < 64; lane++) {
      const int32_t dst = p.grp_dst[g * 64 + lane];
      if (dst < 0) continue;
      double acc = 0.0;
      for (int32_t col = p.grp_off[g]; col < p.grp_off[g + 1]; col++) acc += term(true, col * 64 + lane);
      store(dst, p.grp_aux[g * 64 + lane], acc);
    }
  for (int s = 0; s < p.n_segs(); s++) {
    double acc = 0.0;
    for (int32_t e = p.seg_beg[s]; e < p.seg_end[s]; e++) acc += term(false, e);
    store(p.seg_dst[s], p.seg_aux[s], acc);
  }
  for (int r = 0; r < p.n_blks(); r++) {
    double acc = 0.0;
    for (int32_t e = p.blk_beg[r]; e < p.blk_end[r]; e++) acc += term(false, e);
    store(p.blk_dst[r], p.blk_aux[r], acc);
  }
}

struct HostNet {
  NetworkHost H;
  SparseLU lu;
  SegPlanHost rhs_plan, jac_plan, resid_plan;
  int solve_mode = RES_SOLVE_PLAIN_;
  enum { RES_SOLVE_FUSED_ = 0, RES_SOLVE_EXPLICIT_ = 1, RES_SOLVE_PLAIN_ = 2 };
  std::vector<double> Ea, A;
  int has_kmax = 0;
  double k_max = 0, t_mult = 1.0;
  const std::vector<int32_t>& I(const DevBuf<int32_t>& d) const { return lu.host_i32.at(&d); }
  const std::vector<float>& F(const DevBuf<float>& d) const { return lu.host_f32.at(&d); }
  const SegPlanHost& Pl(const SegPlanDev& d) const { return lu.host_plan.at(&d); }
};

struct HostBackend {
  const HostNet& net;
  const ResParams& P;
  int N, R;
  std::vector<double> k, D, y, psi, d, scale, f0, f1, ytmp, chunk_start, jv, rate, dr, W, jd, sol, sol_t;
  const double* u0 = nullptr;
  const double* k_table = nullptr;
  const double* T_stops = nullptr;
  struct Slot { double c_fact = 0, crate = 1; long long crate_step = 0, crate_restart = -1, last_use = 0, jac_stamp = 0, step_stamp = 0; int valid = 0; };
  std::vector<Slot> slots;
  int64_t n_gj_pivots = 0;

  HostBackend(const HostNet& n, const ResParams& p) : net(n), P(p), N((int)n.H.N), R((int)n.H.R) {
    k.assign(R, 0); D.assign((size_t)RES_D_ROWS * N, 0);
    for (auto* v : {&y, &psi, &d, &scale, &f0, &f1, &ytmp, &chunk_start}) v->assign(N, 0.0);
    jv.assign(net.H.nnz(), 0); rate.assign(R, 0); dr.assign(2 * (size_t)R + 2, 0);
    W.assign((size_t)P.n_slots * net.lu.w_size, 0.0);
    jd.assign((size_t)P.n_slots * N, 0.0);
    sol.assign((size_t)P.sol_cap * N, 0.0); sol_t.assign(P.sol_cap, 0.0);
    slots.assign(P.n_slots, Slot{});
  }
  int n_species() const { return N; }
  void profile_out(int64_t*) const {}
  // slot table
  double slot_c_fact(int i) const { return slots[i].c_fact; }
  double slot_crate(int i) const { return slots[i].crate; }
  long long slot_crate_step(int i) const { return slots[i].crate_step; }
  long long slot_crate_restart(int i) const { return slots[i].crate_restart; }
  void slot_touch(int i, long long c) { slots[i].last_use = c; }
  void slot_rate(int i, double cr, long long st, long long rs) { slots[i].crate = cr; slots[i].crate_step = st; slots[i].crate_restart = rs; }
  void slot_drop(int i) { slots[i].valid = 0; }
  void slot_made(int i, double c, long long clock, long long js, long long ss) {
    Slot& q = slots[i]; q.c_fact = c; q.crate = 1.0; q.valid = 1; q.last_use = clock; q.jac_stamp = js; q.step_stamp = ss;
  }
  void slots_invalidate(bool reset) { for (auto& q : slots) { q.valid = 0; if (reset) { q.c_fact = 0.0; q.last_use = 0; } } }
  int nearest_slot(double c, double band, long long n_restarts, long long max_age) const {
    int best = -1; double bd = 1e300;
    for (int i = 0; i < (int)slots.size(); i++) {
      const Slot& q = slots[i];
      if (!q.valid || n_restarts - q.jac_stamp > max_age) continue;
      const double r = std::fabs(std::log(c / q.c_fact));
      if (r < bd && std::fabs(c / q.c_fact - 1.0) <= band) { bd = r; best = i; }
    }
    return best;
  }
  int victim_slot(long long n_restarts, long long max_age, int n_slots) const {
    for (int i = 0; i < n_slots; i++) if (!slots[i].valid || n_restarts - slots[i].jac_stamp > max_age) return i;
    int v = 0;
    for (int i = 1; i < n_slots; i++) if (slots[i].last_use < slots[v].last_use) v = i;
    return v;
  }
  // vectors
  void load_u0() { std::copy(u0, u0 + N, y.begin()); }
  void chunk_start_from_y() { chunk_start = y; }
  void y_from_chunk_start_clipped() { for (int i = 0; i < N; i++) y[i] = chunk_start[i] < 0.0 ? 0.0 : chunk_start[i]; }
  void y_from_D0() { std::copy(D.begin(), D.begin() + N, y.begin()); }
  void ytmp_from_D0() { std::copy(D.begin(), D.begin() + N, ytmp.begin()); }
  void ytmp_axpy(double h0) { for (int i = 0; i < N; i++) ytmp[i] = y[i] + h0 * f0[i]; }
  void save_y(long long row, double time) { std::copy(y.begin(), y.end(), sol.begin() + (size_t)row * N); sol_t[row] = time; }
  void set_time(long long row, double time) { sol_t[row] = time; }
  void apply_rates(long long stop) {
    if (P.rate_mode == 1) std::copy(k_table + (size_t)stop * R, k_table + (size_t)(stop + 1) * R, k.begin());
    else if (P.rate_mode == 2) {
      const double RT = 8.314462618 * T_stops[stop];
      for (int r = 0; r < R; r++) {
        const double kr = net.A[r] * std::exp(-net.Ea[r] / RT) * 6.02214076e23 * net.t_mult;
        k[r] = net.has_kmax ? 1.0 / (1.0 / net.k_max + 1.0 / kr) : kr;
      }
    }
  }
  void rates(const double* u) {
    for (int r = 0; r < R; r++) {
      const int32_t a = net.H.x0[r], b = net.H.x1[r];
      rate[r] = k[r] * u[a] * (b >= 0 ? u[b] : 1.0);
    }
  }
  void rhs(const double* u, double* out) { rates(u); seg_run_host(SEG_COEF_SET, net.rhs_plan, rate.data(), out, Ex{}); }
  void rhs_y_to_f0() { rhs(y.data(), f0.data()); }
  void rhs_ytmp_to_f1() { rhs(ytmp.data(), f1.data()); }
  void rhs_ytmp_to_f0() { rhs(ytmp.data(), f0.data()); }
  void eval_jac_y() {
    for (int r = 0; r < R; r++) {
      const int32_t a = net.H.x0[r], b = net.H.x1[r];
      double d0, d1 = 0.0;
      if (b < 0) d0 = k[r];
      else if (b == a) d0 = 2.0 * k[r] * y[a];
      else { d0 = k[r] * y[b]; d1 = k[r] * y[a]; }
      dr[2 * r] = d0; dr[2 * r + 1] = d1;
    }
    seg_run_host(SEG_COEF_SET, net.jac_plan, dr.data(), jv.data(), Ex{});
  }
  ResNorms norms(bool with_f1, double atol, double rtol) {
    double s0 = 0, s1 = 0, s2 = 0, vm = 0; int bad = 0;
    for (int i = 0; i < N; i++) {
      const double sc = atol + rtol * std::fabs(y[i]);
      const double a = y[i] / sc, b = f0[i] / sc;
      s0 += a * a; s1 += b * b;
      vm = std::fmax(vm, std::fabs(f0[i]) / (0.1 * std::fabs(y[i]) + sc));
      if (!std::isfinite(f0[i])) bad = 1;
      if (with_f1) { const double c = (f1[i] - f0[i]) / sc; s2 += c * c; if (!std::isfinite(f1[i])) bad = 1; }
    }
    return ResNorms{std::sqrt(s0 / N), std::sqrt(s1 / N), std::sqrt(s2 / N), vm, bad};
  }
  void init_D(bool from_ytmp, double h) {
    const std::vector<double>& y0 = from_ytmp ? ytmp : y;
    std::fill(D.begin(), D.end(), 0.0);
    for (int i = 0; i < N; i++) { D[i] = y0[i]; D[(size_t)N + i] = f0[i] * h; }
  }
  void predict(int order, const double* gamma, double alpha_o, double atol, double rtol) {
    for (int i = 0; i < N; i++) {
      double yp = D[i], ps = 0.0;
      for (int j = 1; j <= order; j++) { const double dj = D[(size_t)j * N + i]; yp += dj; ps += dj * gamma[j]; }
      y[i] = yp; psi[i] = ps / alpha_o; d[i] = 0.0; scale[i] = atol + rtol * std::fabs(yp);
    }
  }
  void change_D(int ord, const double (*RU)[6]) {
    double v[6], o[6];
    for (int i = 0; i < N; i++) {
      for (int j = 0; j <= ord; j++) v[j] = D[(size_t)j * N + i];
      for (int a = 0; a <= ord; a++) { double t = 0.0; for (int q = 0; q <= ord; q++) t += RU[q][a] * v[q]; o[a] = t; }
      for (int j = 0; j <= ord; j++) D[(size_t)j * N + i] = o[j];
    }
  }
  void accept(int order) {
    for (int i = 0; i < N; i++) {
      const double di = d[i];
      D[(size_t)(order + 2) * N + i] = di - D[(size_t)(order + 1) * N + i];
      D[(size_t)(order + 1) * N + i] = di;
      double carry = di;
      for (int j = order; j >= 0; j--) { carry += D[(size_t)j * N + i]; D[(size_t)j * N + i] = carry; }
    }
  }
  void interp(int order, const double* p, long long row) {
    for (int i = 0; i < N; i++) {
      double v = D[i];
      for (int j = 1; j <= order; j++) v += p[j] * D[(size_t)j * N + i];
      sol[(size_t)row * N + i] = v;
    }
  }
  int drift_check(double max_drift) {
    int n = 0;
    for (int s = 0; s < (int)slots.size(); s++) {
      if (!slots[s].valid) continue;
      const double c = slots[s].c_fact;
      double worst = 1.0;
      for (int i = 0; i < N; i++) {
        const double q = (1.0 - c * jd[(size_t)s * N + i]) / (1.0 - c * jv[net.H.j_diag[i]]);
        const double dev = q > 0.0 ? std::max(q, 1.0 / q) : 1e300;
        worst = std::max(worst, dev == dev ? dev : 1e300);
      }
      if (!(worst - 1.0 <= max_drift)) { slots[s].valid = 0; n++; }
    }
    return n;
  }
  double* slot_W(int s) { return W.data() + (size_t)s * net.lu.w_size; }
  bool factor(int slot, double c, bool keep_diag) {
    const SparseLU& lu = net.lu;
    double* Wp = slot_W(slot);
    bool bad = false;
    std::fill(Wp, Wp + lu.off_y, 0.0);
    const auto& jmap = net.I(lu.jmap);
    for (int64_t e = 0; e < lu.nnzJ; e++) { const int32_t jm = jmap[e]; Wp[jm & 0x7fffffff] = (jm < 0 ? 1.0 : 0.0) - c * jv[e]; }
    for (int dd = lu.m; dd < lu.mpad; dd++) Wp[lu.off_S + (int64_t)dd * lu.mpad + dd] = 1.0;
    const auto& ent_pivot = net.I(lu.ent_pivot);
    for (int r = 0; r < lu.nrounds; r++) {
      for (int32_t e = lu.ent_ptr[lu.round_ptr[r]]; e < lu.ent_ptr[lu.round_ptr[r + 1]]; e++) {
        const double w = Wp[lu.off_L + e], piv = Wp[lu.off_diag + ent_pivot[e]];
        const double l = w / piv;
        if (!(std::fabs(piv) >= 1e-8) || (w != 0.0 && !(std::fabs(l) <= 1e8))) bad = true;
        Wp[lu.off_L + e] = l;
      }
      seg_run_host(SEG_PROD_SUB, net.Pl(lu.schur[r]), Wp, Wp, Ex{});
    }
    if (net.solve_mode != HostNet::RES_SOLVE_PLAIN_) {
      for (int i = 0; i < lu.ns; i++) Wp[lu.off_dinv + i] = 1.0 / Wp[lu.off_diag + i];
      const auto& mep = net.I(lu.mono_ent_ptr); const auto& mp = net.I(lu.mono_ptr); const auto& mf = net.I(lu.mono_fac); const auto& md = net.I(lu.mono_dst);
      const auto& ms = net.F(lu.mono_sign);
      std::vector<double> vals(lu.n_mono_ent);
      for (int e = 0; e < lu.n_mono_ent; e++) {
        double acc = 0.0;
        for (int32_t mo = mep[e]; mo < mep[e + 1]; mo++) { double prod = (double)ms[mo]; for (int32_t f = mp[mo]; f < mp[mo + 1]; f++) prod *= Wp[mf[f]]; acc += prod; }
        vals[e] = acc;
      }
      for (int e = 0; e < lu.n_mono_ent; e++) Wp[md[e]] = vals[e];
      if (net.solve_mode == HostNet::RES_SOLVE_FUSED_) {
        seg_run_host(SEG_PROD_AUXSUB, net.Pl(lu.lz_build), Wp, Wp, Ex{});
        seg_run_host(SEG_PROD_NEG, net.Pl(lu.nvu_build), Wp, Wp, Ex{});
      }
    }
    if (lu.m > 0) {   // in-place Gauss-Jordan without pivoting (resident.hip: gj_inplace)
      double* S = Wp + lu.off_S; const int ld = lu.mpad, m = lu.m;
      std::vector<double> row(m), col(m);
      for (int kk = 0; kk < m; kk++) {
        for (int j = 0; j < m; j++) { row[j] = S[(size_t)kk * ld + j]; col[j] = S[(size_t)j * ld + kk]; }
        const double piv = row[kk], p = 1.0 / piv;
        if (!(std::fabs(piv) >= 1e-8)) bad = true;
        for (int i = 0; i < m; i++) {
          const double f = col[i];
          if (i == kk) { for (int j = 0; j < m; j++) S[(size_t)i * ld + j] = j == kk ? p : row[j] * p; }
          else {
            if (f != 0.0 && !(std::fabs(f * p) <= 1e8)) bad = true;
            for (int j = 0; j < m; j++) S[(size_t)i * ld + j] = j == kk ? -f * p : S[(size_t)i * ld + j] - f * (row[j] * p);
          }
        }
      }
    }
    if (keep_diag) for (int i = 0; i < N; i++) jd[(size_t)slot * N + i] = jv[net.H.j_diag[i]];
    return bad;
  }
  void gemv(const double* S, int ld, int m, const double* y2, double* x) {
    std::vector<double> out(m);
    for (int r = 0; r < m; r++) { double acc = 0.0; for (int j = 0; j < m; j++) acc += S[(size_t)r * ld + j] * y2[j]; out[r] = acc; }
    std::copy(out.begin(), out.end(), x);
  }
  void solve(double* Wp) {
    const SparseLU& lu = net.lu;
    const double* Sinv = Wp + lu.off_S;
    const Ex ex{};
    if (net.solve_mode == HostNet::RES_SOLVE_FUSED_) {
      seg_run_host(SEG_PROD_AUXSUB, net.Pl(lu.stageA), Wp, Wp, ex);
      gemv(Sinv, lu.mpad, lu.m, Wp + lu.off_y + lu.ns, Wp + lu.off_x);
      seg_run_host(SEG_PROD_SET, net.Pl(lu.stageC), Wp, Wp, ex);
    } else if (net.solve_mode == HostNet::RES_SOLVE_EXPLICIT_) {
      seg_run_host(SEG_PROD_AUXSUB, net.Pl(lu.fwdZ), Wp, Wp, ex);
      if (lu.m > 0) { seg_run_host(SEG_PROD_SUB, net.Pl(lu.fwd_dense), Wp, Wp, ex); gemv(Sinv, lu.mpad, lu.m, Wp + lu.off_y + lu.ns, Wp + lu.off_x); }
      seg_run_host(SEG_PROD_AUXSUB, net.Pl(lu.bwdT), Wp, Wp, ex);
      seg_run_host(SEG_PROD_SET, net.Pl(lu.bwdV), Wp, Wp, ex);
    } else {
      for (int r = 1; r < lu.nrounds; r++) seg_run_host(SEG_PROD_SUB, net.Pl(lu.fwd[r]), Wp, Wp, ex);
      if (lu.m > 0) {
        if (lu.ns > 0) seg_run_host(SEG_PROD_SUB, net.Pl(lu.fwd_dense), Wp, Wp, ex);
        gemv(Sinv, lu.mpad, lu.m, Wp + lu.off_y + lu.ns, Wp + lu.off_x);
      }
      for (int r = lu.nrounds - 1; r >= 0; r--) seg_run_host(SEG_PROD_SUB_DIV, net.Pl(lu.bwd[r]), Wp, Wp, ex);
    }
  }
  ResAttempt corrector(const ResCorrIn& in, const double* gamma) { return res_corrector_loop(*this, in, gamma, N); }
  void predict_inner(int order, const double* gamma, double alpha_o, double atol, double rtol) { predict(order, gamma, alpha_o, atol, rtol); }
  ResSums newton_iter_inner(int slot, double c, double upd, int order, double ec, double ec_m, double ec_p, double atol, double rtol) {
    double* Wp = slot_W(slot);
    rates(y.data());
    Ex ex; ex.psi = psi.data(); ex.d = d.data(); ex.cscal = c;
    seg_run_host(SEG_COEF_BDF, net.resid_plan, rate.data(), Wp, ex);
    solve(Wp);
    const auto& xloc = net.I(net.lu.xloc);
    ResSums q{0, 0, 0, 0, 0};
    for (int i = 0; i < N; i++) {
      const double dy = upd * Wp[xloc[i]];
      const double s = dy / scale[i];
      q.s += s * s;
      const double yy = y[i] + dy, dd = d[i] + dy;
      const double sce = atol + rtol * std::fabs(yy);
      if (yy < 0.0) q.neg = std::fmax(q.neg, yy < -RES_NEG_DEEP * sce ? RES_NEG_MARK : 1.0);
      const double e = ec * dd / sce;
      q.se += e * e + (std::isfinite(yy) ? 0.0 : INFINITY);
      if (order > 1) { const double em = ec_m * (D[(size_t)order * N + i] + dd) / sce; q.sm += em * em; }
      if (order < RES_MAX_ORDER) { const double ep = ec_p * (dd - D[(size_t)(order + 1) * N + i]) / sce; q.sp += ep * ep; }
      y[i] = yy; d[i] = dd;
    }
    return q;
  }
};

HostNet* make_net(int64_t n_species, int64_t n_reactions, const int64_t* rp, const int64_t* ri, const int64_t* rs, const int64_t* pp,
                  const int64_t* pi, const int64_t* ps, int index_base, const int* lu_opt) {
  HostNet* n = new HostNet();
  n->H = compile_network(n_species, n_reactions, rp, ri, rs, pp, pi, ps, index_base);
  n->lu.host_only = true; n->lu.keep_host = true;
  LUOptions opt;
  opt.min_round = 2;
  if (lu_opt) {
    if (lu_opt[0] > 0) opt.hub_degree = lu_opt[0];
    if (lu_opt[1] > 0) opt.max_rounds = lu_opt[1];
    if (lu_opt[2] > 0) opt.max_tail_degree = lu_opt[2];
    if (lu_opt[3] > 0) opt.max_degree = lu_opt[3];
    if (lu_opt[4] > 0) opt.min_round = lu_opt[4];
  }
  n->lu.analyze((int32_t)n->H.N, n->H.j_ptr, n->H.j_col, opt, nullptr);
  const NetworkHost& H = n->H;
  n->rhs_plan = build_seg_plan(H.N, H.sp_ptr.data(), nullptr, H.sp_rxn.data(), nullptr, H.sp_coef.data(), false);
  n->jac_plan = build_seg_plan(H.nnz(), H.jc_ptr.data(), nullptr, H.jc_src.data(), nullptr, H.jc_coef.data(), false);
  const std::vector<int32_t>& yl = n->I(n->lu.yloc);
  std::vector<int32_t> ident(H.N);
  for (int64_t i = 0; i < H.N; i++) ident[i] = (int32_t)i;
  n->resid_plan = build_seg_plan(H.N, H.sp_ptr.data(), yl.data(), H.sp_rxn.data(), nullptr, H.sp_coef.data(), false, ident.data());
  n->solve_mode = n->lu.fused_tri ? HostNet::RES_SOLVE_FUSED_ : (n->lu.explicit_tri ? HostNet::RES_SOLVE_EXPLICIT_ : HostNet::RES_SOLVE_PLAIN_);
  return n;
}

}  // namespace

extern "C" {

void* res_host_create(int64_t n_species, int64_t n_reactions, const int64_t* rp, const int64_t* ri, const int64_t* rs,
                      const int64_t* pp, const int64_t* pi, const int64_t* ps, int index_base, const int* lu_opt, int64_t* info) {
  try {
    HostNet* n = make_net(n_species, n_reactions, rp, ri, rs, pp, pi, ps, index_base, lu_opt);
    if (info) { info[0] = n->lu.ns; info[1] = n->lu.m; info[2] = n->lu.nrounds; info[3] = n->solve_mode; info[4] = n->lu.w_size; }
    return n;
  } catch (...) { return nullptr; }
}
void res_host_destroy(void* h) { delete (HostNet*)h; }
void res_host_set_arrhenius(void* hv, const double* Ea, const double* A, double k_max, double t_mult) {
  HostNet* n = (HostNet*)hv;
  n->Ea.assign(Ea, Ea + n->H.R); n->A.assign(A, A + n->H.R);
  n->has_kmax = !(k_max != k_max); n->k_max = k_max; n->t_mult = t_mult;
}

// (I - c J(u; k)) x = b through the replayed factorisation: the numerical check of SparseLU::analyze
int res_host_newton_solve(void* hv, double c, const double* k, const double* u, const double* b, double* x) {
  HostNet* n = (HostNet*)hv;
  ResParams P{};
  P.n_slots = 1; P.sol_cap = 1;
  HostBackend B(*n, P);
  std::copy(k, k + n->H.R, B.k.begin());
  std::copy(u, u + n->H.N, B.y.begin());
  B.eval_jac_y();
  const bool bad = B.factor(0, c, false);
  double* W = B.slot_W(0);
  const auto& yl = n->I(n->lu.yloc); const auto& xl = n->I(n->lu.xloc);
  for (int64_t i = 0; i < n->H.N; i++) W[yl[i]] = b[i];
  B.solve(W);
  for (int64_t i = 0; i < n->H.N; i++) x[i] = W[xl[i]];
  return bad ? 1 : 0;
}

// the solve: same arguments as kin_solve (static rates k0 when n_stops == 0); out_t / out_u sized by res_host_rows
int64_t res_host_rows(const kin_params* p) { return make_res_grid(*p).cap; }
int res_host_solve(void* hv, const kin_params* p, const double* u0, const double* k0, const double* tstops, const double* T_stops,
                   const double* k_table, int64_t n_stops, int n_slots, double* out_t, double* out_u, int64_t* n_saved, ResResult* result) {
  HostNet* n = (HostNet*)hv;
  const ResGrid g = make_res_grid(*p);
  ResParams P{};
  res_fill_params(P, *p, g);
  res_default_settings(P, n_slots > 0 ? n_slots : RES_MAX_SLOTS);
  P.save_local = g.save_local.data();
  P.n_stops = (int32_t)n_stops;
  P.rate_mode = n_stops > 0 ? (k_table ? 1 : 2) : 0;
  P.tstops = tstops;
  HostBackend B(*n, P);
  B.u0 = u0; B.k_table = k_table; B.T_stops = T_stops;
  if (n_stops == 0) std::copy(k0, k0 + n->H.R, B.k.begin());
  ResidentBdf<HostBackend> ctl(B, P);
  const ResResult r = ctl.run();
  const int64_t rows = std::min<int64_t>(r.n_saved, g.cap);
  if (out_t) std::copy(B.sol_t.begin(), B.sol_t.begin() + rows, out_t);
  if (out_u) std::copy(B.sol.begin(), B.sol.begin() + (size_t)rows * n->H.N, out_u);
  if (n_saved) *n_saved = rows;
  if (result) *result = r;
  return r.retcode;
}

}  // extern "C"
