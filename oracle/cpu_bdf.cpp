/*
 * cpu_bdf.cpp - CPU baseline / tight-tolerance truth generator for the kinetic ODE solve.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/): tests/, tests/golden/make_*.py,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it - as the checker and as the
 * timed CPU baseline, never as the product. Nothing under kinetica_jl_amd/ links or loads it.
 *
 * What it stands for. The reference integrates with a user-supplied SciML solver; its documented
 * choice is CVODE_BDF(linear_solver=:KLU) (docs/src/getting-started.md:69): a variable-order BDF with
 * modified Newton and KLU's sparse LU, single-threaded. Neither Sundials nor KLU is vendored or
 * present in the image, so this file is a compiled (-O3) CPU implementation of the same class:
 *   - RHS / analytic Jacobian of make_rs (src/solving/solve_utils.jl:318-334; worked ODEs in
 *     docs/src/tutorials/ode-solution.md:33-41), Jacobian assembled in compressed-column form as
 *     ODEProblem(...; jac=true, sparse=true) stores it (methods.jl:157-158);
 *   - sparse LU in KLU's manner: fill-reducing ordering of A + A' (approximate minimum degree,
 *     Amestoy, Davis & Duff 1996, dense rows last), left-looking Gilbert-Peierls factorisation with
 *     threshold partial pivoting that prefers the diagonal (KLU's default tolerance 0.001), and a
 *     pattern-reusing REFACTORISATION for every later matrix (what klu_refactor does for CVODE);
 *   - the integrator of oracle/bdf.py (quasi-constant-step BDF/NDF, orders 1-5), statement for
 *     statement, so that this file, oracle/bdf.py and the device path run the same algorithm;
 *   - the orchestration of src/solving/methods.jl (chunk loop :796-847, save-grid stitching
 *     :829-846, discrete rate updates solve_utils.jl:435-509) and adaptive_solve!
 *     (solve_utils.jl:376-424).
 * PARITY UNPINNED against the reference itself; pinned against oracle/bdf.py + SuperLU
 * (tests/test_cpu_bdf.py) and through it against SciPy's BDF, closed forms and Radau truths.
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double EPS = std::numeric_limits<double>::epsilon();

inline double pow_int(double u, int64_t s) {
  double p = 1.0;
  for (int64_t i = 0; i < s; i++) p *= u;
  return p;
}

// ------------------------------------------------------------------------------------------
// network: flat ragged RxData (network.jl:193-203), 0-based
// ------------------------------------------------------------------------------------------
struct Net {
  int64_t N = 0, R = 0;
  std::vector<int64_t> rp, ri, rs, pp, pi, ps;
  // Jacobian in compressed-column form: pattern + per (reaction, reactant column, affected row) target slot
  std::vector<int32_t> jcp, jri;       // col ptr (N+1), row idx (nnz)
  std::vector<int32_t> jslot;          // one per COO triplet in the order orc-style loops emit them
  int n_threads = 1;

  void rhs(const double* k, const double* u, double* du) const {
    std::memset(du, 0, (size_t)N * sizeof(double));
    for (int64_t r = 0; r < R; r++) {
      double rate = k[r];
      for (int64_t p = rp[r]; p < rp[r + 1]; p++) rate *= pow_int(u[ri[p]], rs[p]);
      for (int64_t p = rp[r]; p < rp[r + 1]; p++) du[ri[p]] -= (double)rs[p] * rate;
      for (int64_t p = pp[r]; p < pp[r + 1]; p++) du[pi[p]] += (double)ps[p] * rate;
    }
  }

  void build_jac_pattern() {
    std::vector<std::pair<int32_t, int32_t>> ent;   // (col, row)
    for (int64_t i = 0; i < N; i++) ent.push_back({(int32_t)i, (int32_t)i});
    for (int64_t r = 0; r < R; r++)
      for (int64_t pj = rp[r]; pj < rp[r + 1]; pj++) {
        for (int64_t p = rp[r]; p < rp[r + 1]; p++) ent.push_back({(int32_t)ri[pj], (int32_t)ri[p]});
        for (int64_t p = pp[r]; p < pp[r + 1]; p++) ent.push_back({(int32_t)ri[pj], (int32_t)pi[p]});
      }
    std::sort(ent.begin(), ent.end());
    ent.erase(std::unique(ent.begin(), ent.end()), ent.end());
    jcp.assign(N + 1, 0);
    jri.resize(ent.size());
    for (size_t e = 0; e < ent.size(); e++) { jcp[ent[e].first + 1]++; jri[e] = ent[e].second; }
    for (int64_t j = 0; j < N; j++) jcp[j + 1] += jcp[j];
    auto find = [&](int32_t col, int32_t row) {
      auto b = jri.begin() + jcp[col], e = jri.begin() + jcp[col + 1];
      return (int32_t)(std::lower_bound(b, e, row) - jri.begin());
    };
    jslot.clear();
    for (int64_t r = 0; r < R; r++)
      for (int64_t pj = rp[r]; pj < rp[r + 1]; pj++) {
        for (int64_t p = rp[r]; p < rp[r + 1]; p++) jslot.push_back(find((int32_t)ri[pj], (int32_t)ri[p]));
        for (int64_t p = pp[r]; p < pp[r + 1]; p++) jslot.push_back(find((int32_t)ri[pj], (int32_t)pi[p]));
      }
  }

  // J[i][j] += nu[i][r] k_r s_jr u_j^(s_jr-1) prod_{l != j} u_l^(s_lr)
  void jac(const double* k, const double* u, double* vals) const {
    std::memset(vals, 0, jri.size() * sizeof(double));
    size_t m = 0;
    for (int64_t r = 0; r < R; r++)
      for (int64_t pj = rp[r]; pj < rp[r + 1]; pj++) {
        const int64_t j = ri[pj], sj = rs[pj];
        double d = k[r] * (double)sj * pow_int(u[j], sj - 1);
        for (int64_t pl = rp[r]; pl < rp[r + 1]; pl++)
          if (pl != pj) d *= pow_int(u[ri[pl]], rs[pl]);
        for (int64_t p = rp[r]; p < rp[r + 1]; p++) vals[jslot[m++]] -= (double)rs[p] * d;
        for (int64_t p = pp[r]; p < pp[r + 1]; p++) vals[jslot[m++]] += (double)ps[p] * d;
      }
  }
};

// ------------------------------------------------------------------------------------------
// approximate minimum degree ordering of the pattern of A + A' (quotient graph, element
// absorption, approximate external degrees; no supervariables). Rows denser than
// max(16, 10 sqrt(n)) are ordered last, as AMD does.
// ------------------------------------------------------------------------------------------
std::vector<int32_t> amd_order(int32_t n, const std::vector<int32_t>& cp, const std::vector<int32_t>& ri) {
  std::vector<std::vector<int32_t>> A(n), E(n), Le(n);
  for (int32_t j = 0; j < n; j++)
    for (int32_t e = cp[j]; e < cp[j + 1]; e++) {
      const int32_t i = ri[e];
      if (i != j) { A[i].push_back(j); A[j].push_back(i); }
    }
  std::vector<char> state(n, 0);   // 0 live variable, 1 eliminated (element), 2 dense (ordered last), 3 dead element
  const int32_t dense = std::max<int32_t>(16, (int32_t)(10.0 * std::sqrt((double)n)));
  std::vector<int32_t> last;
  for (int32_t i = 0; i < n; i++) {
    std::sort(A[i].begin(), A[i].end());
    A[i].erase(std::unique(A[i].begin(), A[i].end()), A[i].end());
  }
  for (int32_t i = 0; i < n; i++)
    if ((int32_t)A[i].size() > dense) { state[i] = 2; last.push_back(i); }
  for (int32_t i = 0; i < n; i++) {
    if (state[i] == 2) { A[i].clear(); continue; }
    auto& a = A[i];
    a.erase(std::remove_if(a.begin(), a.end(), [&](int32_t v) { return state[v] == 2; }), a.end());
  }
  std::vector<int32_t> deg(n, 0), head(n + 1, -1), next(n, -1), prev(n, -1);
  auto bucket_insert = [&](int32_t i) {
    const int32_t d = deg[i];
    next[i] = head[d]; prev[i] = -1;
    if (head[d] >= 0) prev[head[d]] = i;
    head[d] = i;
  };
  auto bucket_remove = [&](int32_t i) {
    const int32_t d = deg[i];
    if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d] = next[i];
    if (next[i] >= 0) prev[next[i]] = prev[i];
  };
  int32_t n_live = 0;
  for (int32_t i = 0; i < n; i++)
    if (state[i] == 0) { deg[i] = (int32_t)A[i].size(); bucket_insert(i); n_live++; }
  std::vector<int32_t> order;
  order.reserve(n);
  std::vector<int64_t> w(n, 0);
  int64_t tag = 1;
  std::vector<int32_t> mark(n, -1), Lp;
  int32_t mind = 0;
  for (int32_t k = 0; k < n_live; k++) {
    while (mind < n && head[mind] < 0) mind++;
    const int32_t p = head[mind];
    bucket_remove(p);
    // Lp = A_p u (u_{e in E_p} L_e) \ {p}
    Lp.clear();
    mark[p] = k;
    for (int32_t v : A[p]) if (state[v] == 0 && mark[v] != k) { mark[v] = k; Lp.push_back(v); }
    for (int32_t e : E[p]) {
      if (state[e] != 1) continue;
      for (int32_t v : Le[e]) if (state[v] == 0 && mark[v] != k) { mark[v] = k; Lp.push_back(v); }
      state[e] = 3;                       // absorbed into the new element p
      std::vector<int32_t>().swap(Le[e]);
    }
    state[p] = 1;
    order.push_back(p);
    std::vector<int32_t>().swap(A[p]);
    std::vector<int32_t>().swap(E[p]);
    // w(e) - tag = |L_e \ Lp| for every element adjacent to a variable of Lp
    tag += n + 1;
    for (int32_t i : Lp) {
      auto& ei = E[i];
      ei.erase(std::remove_if(ei.begin(), ei.end(), [&](int32_t e) { return state[e] != 1 || e == p; }), ei.end());
      for (int32_t e : ei) {
        if (w[e] < tag) w[e] = tag + (int64_t)Le[e].size();
        w[e]--;
      }
    }
    const int32_t lp = (int32_t)Lp.size();
    for (int32_t i : Lp) {
      bucket_remove(i);
      auto& ai = A[i];
      ai.erase(std::remove_if(ai.begin(), ai.end(), [&](int32_t v) { return state[v] != 0 || mark[v] == k; }), ai.end());
      int64_t d = (int64_t)ai.size() + (lp - 1);
      for (int32_t e : E[i]) d += (w[e] - tag);
      d = std::min<int64_t>(d, (int64_t)deg[i] + (lp - 1));
      d = std::min<int64_t>(d, n_live - k - 1);
      E[i].push_back(p);
      deg[i] = (int32_t)std::max<int64_t>(d, 0);
      bucket_insert(i);
      if (deg[i] < mind) mind = deg[i];
    }
    Le[p] = Lp;
  }
  // dense rows last, lowest degree first
  std::sort(last.begin(), last.end());
  for (int32_t i : last) order.push_back(i);
  return order;
}

// ------------------------------------------------------------------------------------------
// sparse LU of a matrix given in compressed-column form (pattern fixed, values change):
//   P (A(q, q)) = L U, q = fill-reducing symmetric ordering, P = partial pivoting inside it
// ------------------------------------------------------------------------------------------
struct SparseLU {
  int32_t n = 0;
  std::vector<int32_t> q, qinv;                // column order / its inverse (symmetric pre-ordering)
  std::vector<int32_t> cp, ri, src;            // permuted pattern: column k of B = A(q, q); src = position in A's values
  // factors: L unit lower (diagonal implicit), U upper with its diagonal stored separately
  std::vector<int64_t> Lp, Up;
  std::vector<int32_t> Li, Ui;                 // row indices in B's row numbering (L) / pivot positions (U)
  std::vector<double> Lx, Ux, Udiag;
  std::vector<int32_t> pinv, prow;             // row -> pivot position, pivot position -> row
  bool have_pattern = false;
  double tol = 1e-3;                           // KLU's default partial-pivoting threshold
  int64_t n_full = 0, n_refactor = 0;
  std::vector<double> x;
  std::vector<int32_t> xi, stack_, pstack;
  std::vector<int32_t> flag;

  void analyze(int32_t n_, const std::vector<int32_t>& acp, const std::vector<int32_t>& ari) {
    n = n_;
    q = amd_order(n, acp, ari);
    qinv.assign(n, 0);
    for (int32_t k = 0; k < n; k++) qinv[q[k]] = k;
    cp.assign(n + 1, 0);
    ri.clear(); src.clear();
    for (int32_t k = 0; k < n; k++) {
      const int32_t j = q[k];
      std::vector<std::pair<int32_t, int32_t>> col;
      for (int32_t e = acp[j]; e < acp[j + 1]; e++) col.push_back({qinv[ari[e]], e});
      std::sort(col.begin(), col.end());
      for (auto& c : col) { ri.push_back(c.first); src.push_back(c.second); }
      cp[k + 1] = (int32_t)ri.size();
    }
    x.assign(n, 0.0); xi.assign(2 * (size_t)n, 0); stack_.assign(n, 0); pstack.assign(n, 0); flag.assign(n, -1);
    have_pattern = false;
  }

  // depth-first search through the graph of L from row i (CSparse's cs_dfs idea): appends the reach in
  // reverse topological order to xi[top..n)
  int32_t dfs(int32_t i, int32_t k, int32_t top) {
    int32_t head = 0;
    stack_[0] = i;
    while (head >= 0) {
      const int32_t r = stack_[head];
      const int32_t jp = pinv[r];
      if (flag[r] != k) { flag[r] = k; pstack[head] = jp < 0 ? 0 : (int32_t)Lp[jp]; }
      bool done = true;
      if (jp >= 0) {
        const int32_t p2 = (int32_t)Lp[jp + 1];
        for (int32_t p = pstack[head]; p < p2; p++) {
          const int32_t r2 = Li[p];
          if (flag[r2] == k) continue;
          pstack[head] = p + 1;
          stack_[++head] = r2;
          done = false;
          break;
        }
      }
      if (done) { head--; xi[--top] = r; }
    }
    return top;
  }

  // full factorisation with pivoting; returns false on a structurally / numerically singular column
  bool factor(const double* avals, double shift_scale_c, bool newton_matrix) {
    (void)shift_scale_c; (void)newton_matrix;
    pinv.assign(n, -1); prow.assign(n, -1);
    Lp.assign(n + 1, 0); Up.assign(n + 1, 0);
    Li.clear(); Lx.clear(); Ui.clear(); Ux.clear(); Udiag.assign(n, 0.0);
    std::fill(flag.begin(), flag.end(), -1);
    for (int32_t k = 0; k < n; k++) {
      Lp[k] = (int64_t)Li.size(); Up[k] = (int64_t)Ui.size();
      int32_t top = n;
      for (int32_t e = cp[k]; e < cp[k + 1]; e++)
        if (flag[ri[e]] != k) top = dfs(ri[e], k, top);
      for (int32_t p = top; p < n; p++) x[xi[p]] = 0.0;
      for (int32_t e = cp[k]; e < cp[k + 1]; e++) x[ri[e]] = avals[src[e]];
      // x = L \ B(:, k) in topological order
      for (int32_t p = top; p < n; p++) {
        const int32_t r = xi[p], jp = pinv[r];
        if (jp < 0) continue;
        const double xj = x[r];
        for (int64_t t = Lp[jp]; t < Lp[jp + 1]; t++) x[Li[t]] -= Lx[t] * xj;
      }
      // pivot: largest magnitude among the non-pivotal rows, the diagonal if it is within tol of it
      double amax = -1.0; int32_t ipiv = -1;
      for (int32_t p = top; p < n; p++) {
        const int32_t r = xi[p];
        if (pinv[r] < 0) { const double a = std::fabs(x[r]); if (a > amax) { amax = a; ipiv = r; } }
        else { Ui.push_back(pinv[r]); Ux.push_back(x[r]); }
      }
      if (ipiv < 0 || !(amax > 0.0) || !std::isfinite(amax)) return false;
      if (pinv[k] < 0 && flag[k] == k && std::fabs(x[k]) >= tol * amax) ipiv = k;
      const double piv = x[ipiv];
      Udiag[k] = piv;
      pinv[ipiv] = k; prow[k] = ipiv;
      for (int32_t p = top; p < n; p++) {
        const int32_t r = xi[p];
        if (pinv[r] < 0) { Li.push_back(r); Lx.push_back(x[r] / piv); }
      }
    }
    Lp[n] = (int64_t)Li.size(); Up[n] = (int64_t)Ui.size();
    // U's entries of a column were pushed in topological order of their pivots: the order refactor() replays
    have_pattern = true;
    n_full++;
    return true;
  }

  // same pattern, same pivots, new values (klu_refactor): no search, no pivoting
  bool refactor(const double* avals) {
    if (!have_pattern) return factor(avals, 0.0, false);
    for (int32_t k = 0; k < n; k++) {
      for (int64_t t = Up[k]; t < Up[k + 1]; t++) x[prow[Ui[t]]] = 0.0;
      x[prow[k]] = 0.0;
      for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) x[Li[t]] = 0.0;
      for (int32_t e = cp[k]; e < cp[k + 1]; e++) x[ri[e]] = avals[src[e]];
      for (int64_t t = Up[k]; t < Up[k + 1]; t++) {
        const int32_t jp = Ui[t];
        const double xj = x[prow[jp]];
        Ux[t] = xj;
        for (int64_t s = Lp[jp]; s < Lp[jp + 1]; s++) x[Li[s]] -= Lx[s] * xj;
      }
      const double piv = x[prow[k]];
      // a pivot that lost its dominance: redo the pivoting factorisation (what CVODE's KLU interface does when
      // the refactorisation's condition estimate degrades)
      double amax = std::fabs(piv);
      for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) amax = std::max(amax, std::fabs(x[Li[t]]));
      if (!(std::fabs(piv) >= 1e-3 * tol * amax) || !std::isfinite(piv) || piv == 0.0) return factor(avals, 0.0, false);
      Udiag[k] = piv;
      for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) Lx[t] = x[Li[t]] / piv;
    }
    n_refactor++;
    return true;
  }

  // solves A z = b; b and z in the caller's (unpermuted) numbering
  void solve(const double* b, double* z, std::vector<double>& work) const {
    work.resize(n);
    double* y = work.data();            // indexed by B's row numbering first, by pivot position after the L solve
    std::vector<double>& xx = const_cast<std::vector<double>&>(x);
    for (int32_t i = 0; i < n; i++) xx[qinv[i]] = b[i];      // rows of B = A(q, q)
    // L y = P b : process pivots in order; y_k = xx[prow[k]]
    for (int32_t k = 0; k < n; k++) {
      const double yk = xx[prow[k]];
      y[k] = yk;
      if (yk != 0.0)
        for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) xx[Li[t]] -= Lx[t] * yk;
    }
    // U w = y (columns): w_k = y_k / U_kk, then y -= U(:, k) w_k
    for (int32_t k = n - 1; k >= 0; k--) {
      const double wk = y[k] / Udiag[k];
      y[k] = wk;
      if (wk != 0.0)
        for (int64_t t = Up[k]; t < Up[k + 1]; t++) y[Ui[t]] -= Ux[t] * wk;
    }
    for (int32_t k = 0; k < n; k++) z[q[k]] = y[k];
  }
  int64_t nnz() const { return (int64_t)Li.size() + (int64_t)Ui.size() + n; }
};

// ------------------------------------------------------------------------------------------
// BDF integrator: oracle/bdf.py (OracleBDF), statement for statement
// ------------------------------------------------------------------------------------------
constexpr int MAX_ORDER = 5, NEWTON_MAXITER = 4;
constexpr double MIN_FACTOR = 0.2, MAX_FACTOR = 10.0, FIRST_MAX_FACTOR = 1e4;   // growth cap: first selection after a (re)initialisation / later (CVODE: ETAMX1 / ETAMX2)
const double KAPPA[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};

struct Stats {
  int64_t n_steps = 0, n_rejected = 0, n_rhs = 0, n_jac = 0, n_factor = 0, n_linsolve = 0, n_newton_fail = 0;
  int64_t n_chunks = 0, n_restarts = 0, n_retries = 0, n_resets = 0;
  double final_abstol = 0, final_reltol = 0, wall_seconds = 0;
  int64_t lu_nnz = 0, lu_full = 0, lu_refactor = 0;
  double t_rhs = 0, t_jac = 0, t_factor = 0, t_solve = 0;
};

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

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Bdf {
  const Net& net;
  SparseLU lu;
  int64_t N;
  const double* k = nullptr;            // rate constants in force
  double atol = 0, rtol = 0, newton_tol = 0, dtmin = 0;
  bool ban_negatives = false;
  double lu_reuse = 0.0;                // > 0: keep the factorisation while |c/c_fact - 1| <= lu_reuse (CVODE's DGMAX rule)
  double step_thresh = 0.0;             // > 0: a proposed step-size factor below it leaves h and the order alone (CVODE's THRESH)
  // LU cache (mirrors solver.cpp): factorisations are kept in slots keyed by the bucket of c on a geometric grid
  // (4 per decade) and reused - across step-size changes AND across restarts - while |c / c_fact - 1| <= lu_reuse,
  // with the update scaled by 2 / (1 + c / c_fact); a slot is refreshed (new Jacobian, new factorisation) only when a
  // corrector that used it fails. lu_cache = number of slots (0: off).
  double lu_rate_max = 0.15;  // slowest contraction accepted from a reused factorisation
  bool force_fresh_lu = false, slot_is_fresh = true;
  int64_t steps_since_jac = 0;
  int lu_cache = 0;
  struct Slot { double c_fact = 0.0, crate = 1.0; int64_t crate_step = 0, crate_restart = -1; int64_t last_use = 0, jac_stamp = 0; std::vector<double> Lx, Ux, Udiag, jd; };
  int64_t lu_max_age = 50, jac_stamp_now = 0;
  double lu_drift_max = 1.0;                    // drift guard: see Solver::restart (solver.cpp); 0.25 until round 4   // restarts a slot stays on offer after its Jacobian was evaluated
  bool cache_suspended = false;                // a tolerance retry runs its chunk without the cache
  std::vector<Slot> slots;          // the value arrays of the ACTIVE slot live in `lu` (swapped in)
  int active_slot = -1;
  int64_t use_clock = 0, n_cache_hits = 0;
  void swap_arrays(Slot& s) { s.Lx.swap(lu.Lx); s.Ux.swap(lu.Ux); s.Udiag.swap(lu.Udiag); }
  int nearest_slot(double c) const {      // the slot whose c_fact is closest to c in ratio, -1 if none within the band
    int best = -1; double bd = 1e300;
    for (int i = 0; i < (int)slots.size(); i++) {
      if (st.n_restarts - slots[i].jac_stamp > lu_max_age) continue;
      const double r = std::fabs(std::log(c / slots[i].c_fact));
      if (r < bd && std::fabs(c / slots[i].c_fact - 1.0) <= lu_reuse) { bd = r; best = i; }
    }
    return best;
  }
  void activate(int i) {
    if (i == active_slot) return;
    if (active_slot >= 0) swap_arrays(slots[active_slot]);
    swap_arrays(slots[i]);
    active_slot = i;
  }
  int new_slot() {                        // a free slot, else the least recently used one
    if (active_slot >= 0) swap_arrays(slots[active_slot]);
    active_slot = -1;
    int i = -1;
    for (int j = 0; j < (int)slots.size() && i < 0; j++) if (st.n_restarts - slots[j].jac_stamp > lu_max_age) i = j;
    if (i >= 0) {}
    else if ((int)slots.size() < lu_cache) { slots.emplace_back(); i = (int)slots.size() - 1; }
    else { i = 0; for (int j = 1; j < (int)slots.size(); j++) if (slots[j].last_use < slots[i].last_use) i = j; }
    swap_arrays(slots[i]);
    // working arrays of the right size (refactor overwrites every entry; the first, pivoting factorisation sizes them)
    if (lu.have_pattern) { lu.Lx.resize(lu.Li.size()); lu.Ux.resize(lu.Ui.size()); lu.Udiag.resize(N); }
    active_slot = i;
    return i;
  }
  // a re-pivoted factorisation changes the pattern: every other slot holds values of the old one
  void keep_only_active() {
    Slot mine = slots[active_slot];
    slots.clear(); slots.push_back(mine); active_slot = 0;
  }
  void clear_cache() { slots.clear(); active_slot = -1; }
  std::vector<double> jac_diag() const {
    std::vector<double> jd(N);
    for (int64_t j = 0; j < N; j++) {
      auto b = net.jri.begin() + net.jcp[j], e = net.jri.begin() + net.jcp[j + 1];
      jd[j] = J[std::lower_bound(b, e, (int32_t)j) - net.jri.begin()];
    }
    return jd;
  }
  // drops the slots whose diag(I - c_s J) moved by more than lu_drift_max against the Jacobian just evaluated
  void drift_check() {
    if (lu_cache <= 0 || lu_drift_max <= 0.0 || slots.empty()) return;
    const std::vector<double> now = jac_diag();
    if (active_slot >= 0) { swap_arrays(slots[active_slot]); active_slot = -1; }
    std::vector<Slot> keep;
    for (Slot& sl : slots) {
      double worst = 1.0;
      for (int64_t i = 0; i < N; i++) {
        const double q = (1.0 - sl.c_fact * sl.jd[i]) / (1.0 - sl.c_fact * now[i]);
        const double dev = q > 0.0 ? std::max(q, 1.0 / q) : 1e300;
        worst = std::max(worst, dev == dev ? dev : 1e300);
      }
      if (worst - 1.0 <= lu_drift_max) keep.push_back(std::move(sl));
    }
    slots.swap(keep);
  }
  double GAMMA[7], ALPHA[7], ERRC[7];
  std::vector<double> D, y, ypred, psi, d, scale, f, rhsv, dy, J, M, work, tmp;
  double t = 0, h_abs = 0, c_fact = 0, cur_crate = 1.0;
  // carried convergence rate: validity of the first-iteration test (Solver::crate_fresh, solver.cpp) and the cache-less mode's copy
  bool crate_is_fresh = false;
  int64_t crate_max_age = 10, nc_crate_step = 0, nc_crate_restart = -1;
  double crate_dy_max = 0.2;
  bool crate_fresh(double crate, int64_t step, int64_t restart) const {
    return crate < 1.0 && restart == st.n_restarts && st.n_steps - step <= crate_max_age;
  }
  int order = 1, n_equal = 0;
  bool lu_valid = false, jac_current = false, have_pending = false;
  double pend[4] = {0, 0, 0, 0};
  double fail_score = 0;
  int64_t iters_left = 0;
  Stats st;

  explicit Bdf(const Net& n_) : net(n_), N(n_.N) {
    GAMMA[0] = 0.0;
    for (int j = 1; j <= MAX_ORDER; j++) GAMMA[j] = GAMMA[j - 1] + 1.0 / j;
    for (int j = 0; j <= MAX_ORDER; j++) ALPHA[j] = (1.0 - KAPPA[j]) * GAMMA[j];
    for (int j = 0; j <= MAX_ORDER; j++) ERRC[j] = KAPPA[j] * GAMMA[j] + 1.0 / (j + 1);
    ERRC[MAX_ORDER + 1] = 0.0;
    D.assign((size_t)(MAX_ORDER + 3) * N, 0.0);
    for (auto* v : {&y, &ypred, &psi, &d, &scale, &f, &rhsv, &dy, &tmp}) v->assign(N, 0.0);
    J.assign(net.jri.size(), 0.0); M.assign(net.jri.size(), 0.0);
    lu.analyze((int32_t)N, net.jcp, net.jri);
  }
  double* Drow(int j) { return D.data() + (size_t)j * N; }
  void set_tols(double a, double r) { atol = a; rtol = r; newton_tol = std::max(10.0 * EPS / r, std::min(0.1, std::max(0.03, 1e-10 / r))); }   // oracle/bdf.py set_tols
  void fun(const double* u, double* out) { const double t0 = now_s(); net.rhs(k, u, out); st.n_rhs++; st.t_rhs += now_s() - t0; }
  void eval_jac(const double* u) { const double t0 = now_s(); net.jac(k, u, J.data()); st.n_jac++; lu_valid = false; steps_since_jac = 0; jac_stamp_now = st.n_restarts; st.t_jac += now_s() - t0; }
  double rms_scaled(const double* v, const double* sc) const {
    double s = 0.0;
    for (int64_t i = 0; i < N; i++) { const double q = v[i] / sc[i]; s += q * q; }
    return std::sqrt(s / (double)N);
  }
  bool finite(const double* v) const {
    for (int64_t i = 0; i < N; i++) if (!std::isfinite(v[i])) return false;
    return true;
  }
  void change_D(int ord, double factor) {
    double R[6][6], U[6][6], RU[6][6];
    compute_R(ord, factor, R);
    compute_R(ord, 1.0, U);
    for (int a = 0; a <= ord; a++)
      for (int b = 0; b <= ord; b++) {
        double v = 0.0;
        for (int q = 0; q <= ord; q++) v += R[a][q] * U[q][b];
        RU[a][b] = v;
      }
    double v[6], o[6];
    for (int64_t i = 0; i < N; i++) {
      for (int j = 0; j <= ord; j++) v[j] = D[(size_t)j * N + i];
      for (int a = 0; a <= ord; a++) {
        double tt = 0.0;
        for (int b = 0; b <= ord; b++) tt += RU[b][a] * v[b];
        o[a] = tt;
      }
      for (int j = 0; j <= ord; j++) D[(size_t)j * N + i] = o[j];
    }
  }
  bool factor(double c) {
    const double t0 = now_s();
    const size_t nnz = J.size();
    for (size_t e = 0; e < nnz; e++) M[e] = -c * J[e];
    for (int64_t j = 0; j < N; j++) {   // + I on the diagonal
      auto b = net.jri.begin() + net.jcp[j], e = net.jri.begin() + net.jcp[j + 1];
      M[std::lower_bound(b, e, (int32_t)j) - net.jri.begin()] += 1.0;
    }
    const bool ok = lu.refactor(M.data());
    st.n_factor++;
    if (getenv("CPUB_TRACE")) fprintf(stderr, "[factor] t=%.3e h=%.3e order=%d c=%.3e c_prev=%.3e steps=%lld\n", t, h_abs, order, c, c_fact, (long long)st.n_steps);
    c_fact = c;
    cur_crate = 1.0; nc_crate_step = 0; nc_crate_restart = -1;
    st.t_factor += now_s() - t0;
    return ok;
  }
  bool restart(double t0, const double* y0, double t_bound) {
    st.n_restarts++;
    t = t0;
    std::copy(y0, y0 + N, y.begin());
    fun(y.data(), f.data());
    if (!finite(f.data())) return false;
    const double interval = std::fabs(t_bound - t0);
    for (int64_t i = 0; i < N; i++) scale[i] = atol + std::fabs(y[i]) * rtol;
    // CVODE's initial step (cvhin below), rounded DOWN to a power of ten by exact IEEE operations (kinetica_jl_amd/csrc/solver.cpp:
    // decade_floor - every restart's climb passes through the same step sizes, which is what the LU cache lives on)
    {
      const double hx = cvhin(std::max(std::fabs(t0), std::fabs(t_bound)), interval);
      if (!(hx > 0.0)) return false;
      double p = 1.0;
      while (p > hx) p /= 10.0;
      while (p * 10.0 <= hx) p *= 10.0;
      h_abs = std::isfinite(hx) ? p : hx;
    }
    first_selection = true;
    std::fill(D.begin(), D.end(), 0.0);
    for (int64_t i = 0; i < N; i++) { D[i] = y[i]; D[(size_t)N + i] = f[i] * h_abs; }
    order = 1; n_equal = 0;
    eval_jac(y.data());
    drift_check();
    jac_current = true; have_pending = false; fail_score = 0.0;
    return true;
  }
  // CVODE's initial step (cvode.c: cvHin / cvUpperBoundH0 / cvYddNorm - the solver the reference documents,
  // docs/src/getting-started.md:69, re-initialised at every chunk start and rate update, src/solving/methods.jl:260, 819): the
  // step h with ||h^2 y'' / 2||_WRMS = 1, y'' from a difference quotient of f along the Euler direction, iterated (at most 4
  // evaluations) until two successive estimates agree within a factor of 2, halved (H_BIAS), and kept inside [hlb, hub]:
  // hlb = 100 ulp of the segment's times, hub = a tenth of the segment but no step over which ANY component would move by more
  // than a tenth of itself plus its error weight. y, f = f(y) and scale are current. Returns 0 when f is not finite on the way.
  // The step-size selection that follows the first steps may grow the step by 1e4 (CVODE's ETAMX1), later ones by 10.
  bool first_selection = false;
  double cvhin(double tmax, double tdist) {
    const double hlb = 100.0 * EPS * tmax;
    double hub_inv = 0.0;
    for (int64_t i = 0; i < N; i++) hub_inv = std::max(hub_inv, std::fabs(f[i]) / (0.1 * std::fabs(y[i]) + scale[i]));
    double hub = 0.1 * tdist;
    if (hub * hub_inv > 1.0) hub = 1.0 / hub_inv;
    double hg = std::sqrt(hlb * hub), hnew = hg;
    if (hub >= hlb) {
      for (int count = 1; count <= 4; count++) {
        for (int64_t i = 0; i < N; i++) tmp[i] = y[i] + hg * f[i];
        fun(tmp.data(), rhsv.data());
        if (!finite(rhsv.data())) return 0.0;
        for (int64_t i = 0; i < N; i++) tmp[i] = rhsv[i] - f[i];
        const double ydd = rms_scaled(tmp.data(), scale.data()) / hg;
        hnew = (ydd * hub * hub > 2.0) ? std::sqrt(2.0 / ydd) : std::sqrt(hg * hub);
        if (count == 4) break;
        const double hrat = hnew / hg;
        if (hrat > 0.5 && hrat < 2.0) break;
        if (count > 1 && hrat > 2.0) { hnew = hg; break; }
        hg = hnew;
      }
    }
    double h0 = 0.5 * hnew;
    if (h0 < hlb) h0 = hlb;
    if (h0 > hub) h0 = hub;
    return std::min(h0, tdist);
  }
  void reset_history() {
    std::copy(D.begin(), D.begin() + N, tmp.begin());
    fun(tmp.data(), f.data());
    std::fill(D.begin(), D.end(), 0.0);
    for (int64_t i = 0; i < N; i++) { D[i] = tmp[i]; D[(size_t)N + i] = f[i] * h_abs; }
    order = 1; n_equal = 0; lu_valid = false; fail_score = 0.0; st.n_resets++;
    clear_cache();   // three failed attempts in a row: nothing cached is trusted any more either
  }
  // returns converged; n_iter out
  bool newton(double c, int& n_iter) {
    std::fill(d.begin(), d.end(), 0.0);
    std::copy(ypred.begin(), ypred.end(), y.begin());
    double dy_norm_old = -1.0;
    bool converged = false;
    int kk = 0;
    // a factorisation made for another c is still used (lu_reuse): the update is scaled by 2 / (1 + c/c_fact), the
    // first-order correction CVODE applies for the changed gamma
    const double upd = (lu_reuse > 0.0 && c_fact != c) ? 2.0 / (1.0 + c / c_fact) : 1.0;
    // CVODE's carried convergence rate (cvNlsConvTest: crate <- max(0.3 crate, del / delp), reset to 1 by every setup):
    // every factorisation keeps the contraction its iterations have shown; the first iteration of a step is judged with it
    const double crate0 = cur_crate;
    double crate = cur_crate;
    for (kk = 0; kk < NEWTON_MAXITER; kk++) {
      fun(y.data(), f.data());
      for (int64_t i = 0; i < N; i++) rhsv[i] = c * f[i] - psi[i] - d[i];
      const double t0 = now_s();
      lu.solve(rhsv.data(), dy.data(), work);
      st.n_linsolve++;
      st.t_solve += now_s() - t0;
      if (upd != 1.0) for (int64_t i = 0; i < N; i++) dy[i] *= upd;
      if (!finite(dy.data())) break;
      const double dy_norm = rms_scaled(dy.data(), scale.data());
      const bool have_rate = dy_norm_old >= 0.0;
      const double rate = have_rate ? dy_norm / dy_norm_old : 0.0;
      if (have_rate && std::isfinite(dy_norm)) crate = std::max(0.3 * crate, rate);
      const double rate_max = (lu_cache > 0 && !cache_suspended && !slot_is_fresh) ? lu_rate_max : 1.0;
      if (have_rate && (rate >= rate_max || std::pow(rate, NEWTON_MAXITER - kk) / (1.0 - rate) * dy_norm > newton_tol)) break;
      for (int64_t i = 0; i < N; i++) { y[i] += dy[i]; d[i] += dy[i]; }
      if (dy_norm == 0.0 || (have_rate && rate / (1.0 - rate) * dy_norm < newton_tol) ||
          (!have_rate && (dy_norm < newton_tol || (crate_is_fresh && crate0 < 1.0 && dy_norm <= crate_dy_max &&
                                                   crate0 / (1.0 - crate0) * dy_norm < newton_tol)))) {
        converged = true;
        break;
      }
      dy_norm_old = dy_norm;
    }
    cur_crate = crate;
    n_iter = std::min(kk + 1, NEWTON_MAXITER);
    return converged;
  }
  // corrector with the LU cache: use the slot of c's bucket when it is close enough in c; a failure with a slot that was
  // not made for this very matrix refreshes it (Jacobian at the predictor + factorisation at this c) and retries once
  bool corrector_cached(double c, int& n_iter) {
    bool fresh = false;
    int i = nearest_slot(c);
    if (i >= 0 && !force_fresh_lu) {
      activate(i);
      n_cache_hits++;
    } else {
      if (force_fresh_lu && !jac_current && steps_since_jac > 20) { eval_jac(ypred.data()); jac_current = true; }
      if (i >= 0) activate(i); else i = new_slot();
      const int64_t pb = lu.n_full;
      if (!factor(c)) return false;
      if (lu.n_full != pb && pb != 0) { keep_only_active(); i = 0; }
      slots[i].c_fact = c; slots[i].crate = 1.0; slots[i].crate_step = 0; slots[i].crate_restart = -1; slots[i].jac_stamp = jac_stamp_now; slots[i].jd = jac_diag(); fresh = jac_current;
    }
    force_fresh_lu = false;
    slots[i].last_use = ++use_clock;
    c_fact = slots[i].c_fact;
    slot_is_fresh = fresh || c_fact == c;
    cur_crate = slots[i].crate;
    crate_is_fresh = crate_fresh(slots[i].crate, slots[i].crate_step, slots[i].crate_restart);
    bool converged = newton(c, n_iter);
    if (n_iter > 1) { slots[i].crate = cur_crate; slots[i].crate_step = st.n_steps; slots[i].crate_restart = st.n_restarts; }
    if (converged) {
      if (!fresh && n_iter >= NEWTON_MAXITER) {   // too stale to be offered again
        if (active_slot >= 0) swap_arrays(slots[active_slot]);
        slots.erase(slots.begin() + i);
        active_slot = -1;
      }
      return true;
    }
    st.n_newton_fail++;
    if (fresh) return false;            // current Jacobian, matrix made for this c: the step itself is too long
    if (!jac_current) { eval_jac(ypred.data()); jac_current = true; }
    const int64_t pb = lu.n_full;
    if (!factor(c)) return false;
    if (lu.n_full != pb) { keep_only_active(); i = 0; }
    slots[i].c_fact = c; slots[i].last_use = ++use_clock; slots[i].jac_stamp = jac_stamp_now; slots[i].jd = jac_diag();
    c_fact = c;
    slot_is_fresh = true;
    cur_crate = 1.0;
    crate_is_fresh = false;
    slots[i].crate = 1.0; slots[i].crate_step = 0; slots[i].crate_restart = -1;
    converged = newton(c, n_iter);
    if (n_iter > 1) { slots[i].crate = cur_crate; slots[i].crate_step = st.n_steps; slots[i].crate_restart = st.n_restarts; }
    if (!converged) st.n_newton_fail++;
    return converged;
  }
  enum Status { OK = 0, DTMIN = 1, MAXITERS = 2, UNSTABLE = 3 };
  static constexpr double NEG_DEEP = 1e3;   // kinetica_jl_amd/csrc/solver_kernels.hpp BDF_NEG_DEEP
  Status step(double t_bound) {
    bool accepted = false, first_attempt = true;
    double safety = 0.9, err_norm = 0.0, t_new = t;
    while (!accepted) {
      iters_left--;
      if (iters_left < 0) return MAXITERS;
      const double min_step = std::max(dtmin, 10.0 * (std::nextafter(t, INF) - t));
      if (h_abs < min_step) {
        if (!first_attempt) return DTMIN;
        change_D(order, min_step / h_abs);
        h_abs = min_step; n_equal = 0; lu_valid = false;
      }
      first_attempt = false;
      t_new = t + h_abs;
      if (t_new - t_bound > 0.0) {
        t_new = t_bound;
        change_D(order, std::fabs(t_new - t) / h_abs);
        n_equal = 0; lu_valid = false;
      }
      const double h = t_new - t;
      h_abs = std::fabs(h);
      for (int64_t i = 0; i < N; i++) {
        double yp = D[i], ps = 0.0;
        for (int j = 1; j <= order; j++) { const double dj = D[(size_t)j * N + i]; yp += dj; ps += dj * GAMMA[j]; }
        ypred[i] = yp; psi[i] = ps / ALPHA[order]; scale[i] = atol + rtol * std::fabs(yp);
      }
      const double c = h / ALPHA[order];
      bool converged = false;
      int n_iter = 0;
      if (lu_cache > 0 && !cache_suspended) {
        converged = corrector_cached(c, n_iter);
      } else
      for (;;) {
        const bool reusable = lu_reuse > 0.0 && lu.have_pattern && c_fact != 0.0 && !lu_stale &&
                              std::fabs(c / c_fact - 1.0) <= lu_reuse;
        if (!lu_valid && !reusable) {
          if (!factor(c)) { converged = false; break; }
          lu_valid = true; lu_stale = false;
        }
        crate_is_fresh = crate_fresh(cur_crate, nc_crate_step, nc_crate_restart);
        converged = newton(c, n_iter);
        if (n_iter > 1) { nc_crate_step = st.n_steps; nc_crate_restart = st.n_restarts; }
        if (converged) break;
        st.n_newton_fail++;
        if (lu_reuse > 0.0 && c_fact != c) {   // the stale factorisation is the first suspect: refactor for this c, same Jacobian
          lu_valid = false; lu_stale = true;
          continue;
        }
        if (jac_current) break;
        eval_jac(ypred.data());
        jac_current = true; lu_stale = true;
      }
      bool negative = false;
      if (converged && ban_negatives)
        for (int64_t i = 0; i < N; i++) if (y[i] < 0.0) { negative = true; break; }
      if (!converged || negative) {
        // a failed corrector cuts the step to a quarter and does not count towards the history reset (CVODE: ETACF = 0.25,
        // history rebuilt only after repeated error-test failures); a banned negative state halves it and counts
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
      safety = 0.9 * (2.0 * NEWTON_MAXITER + 1.0) / (2.0 * NEWTON_MAXITER + n_iter);
      for (int64_t i = 0; i < N; i++) scale[i] = atol + rtol * std::fabs(y[i]);
      for (int64_t i = 0; i < N; i++) tmp[i] = ERRC[order] * d[i];
      err_norm = rms_scaled(tmp.data(), scale.data());
      if (err_norm > 1.0) {
        const double factor = std::max(MIN_FACTOR, safety * std::pow(err_norm, -1.0 / (order + 1)));
        h_abs *= factor;
        change_D(order, factor);
        n_equal = 0;
        force_fresh_lu = lu_cache > 0 && !cache_suspended;
        st.n_rejected++;
        first_selection = false;
        fail_score += 1.0;
        if (fail_score >= 3.0 && order > 1) reset_history();
      } else {
        // an accepted step with a species below -NEG_DEEP error weights: the negative excursion, given up early
        for (int64_t i = 0; i < N; i++) if (y[i] < -NEG_DEEP * scale[i]) return UNSTABLE;
        accepted = true;
      }
    }
    st.n_steps++;
    steps_since_jac++;
    fail_score = std::max(0.0, fail_score - 0.2);
    n_equal++;
    t = t_new;
    for (int64_t i = 0; i < N; i++) {
      const double di = d[i];
      D[(size_t)(order + 2) * N + i] = di - D[(size_t)(order + 1) * N + i];
      D[(size_t)(order + 1) * N + i] = di;
      double carry = di;
      for (int j = order; j >= 0; j--) { carry += D[(size_t)j * N + i]; D[(size_t)j * N + i] = carry; }
    }
    jac_current = false;
    have_pending = false;
    if (n_equal >= order + 1) {
      double em = INF, ep = INF;
      if (order > 1) {
        for (int64_t i = 0; i < N; i++) tmp[i] = ERRC[order - 1] * D[(size_t)order * N + i];
        em = rms_scaled(tmp.data(), scale.data());
      }
      if (order < MAX_ORDER) {
        for (int64_t i = 0; i < N; i++) tmp[i] = ERRC[order + 1] * D[(size_t)(order + 2) * N + i];
        ep = rms_scaled(tmp.data(), scale.data());
      }
      pend[0] = em; pend[1] = err_norm; pend[2] = ep; pend[3] = safety;
      have_pending = true;
    }
    return OK;
  }
  bool lu_stale = false;   // the factorisation in hand failed a corrector: do not reuse it for a neighbouring c
  void select_order() {
    if (!have_pending) return;
    have_pending = false;
    double best = -1.0;
    int arg = 1;
    for (int i = 0; i < 3; i++) {
      double fct;
      if (pend[i] == 0.0) fct = INF;
      else if (std::isinf(pend[i])) fct = 0.0;
      else fct = std::pow(pend[i], -1.0 / (order + i));
      if (fct > best) { best = fct; arg = i; }
    }
    const double factor = std::min(first_selection ? FIRST_MAX_FACTOR : MAX_FACTOR, pend[3] * best);
    first_selection = false;
    if (step_thresh > 0.0 && factor < step_thresh) return;   // not worth a new iteration matrix: looked at again next step
    order += arg - 1;
    h_abs *= factor;
    change_D(order, factor);
    n_equal = 0;
    lu_valid = false;
  }
  void interpolate(double ts, double* out) {
    double p[6];
    double prod = 1.0;
    for (int j = 0; j < order; j++) { prod *= (ts - (t - h_abs * j)) / (h_abs * (1.0 + j)); p[j + 1] = prod; }
    for (int64_t i = 0; i < N; i++) {
      double v = D[i];
      for (int j = 1; j <= order; j++) v += p[j] * D[(size_t)j * N + i];
      out[i] = v;
    }
  }
};

struct CpuParams {
  double tspan0, tspan1, abstol, reltol;
  int32_t adaptive_tols, solve_chunks, ban_negatives, reserved;
  double solve_chunkstep;
  int64_t maxiters;
  double save_interval;   // < 0: nothing
  double dtmin;           // <= 0: eps(chunkstep) / eps(tspan1)
  double lu_reuse;        // 0: refactor at every change of c (oracle/bdf.py's default); > 0: CVODE-style reuse band
  double step_thresh;     // 0: every proposed step-size change is taken; > 0: CVODE's THRESH
  int64_t lu_cache;       // 0: one factorisation in hand; > 0: that many cached factorisations, reused across restarts
};

struct Handle {
  Net net;
  Bdf* bdf = nullptr;
  std::vector<double> sol_t, sol_u;
  Stats st;
  ~Handle() { delete bdf; }
};

enum { RET_SUCCESS = 0, RET_MAXITERS = 1, RET_DTMIN = 2, RET_UNSTABLE = 3 };

int solve(Handle& H, const CpuParams& p, const double* u0, const double* k0, const double* tstops, const double* k_table,
          int64_t n_stops) {
  const double wall0 = now_s();
  const int64_t N = H.net.N, R = H.net.R;
  if (!H.bdf) H.bdf = new Bdf(H.net);
  Bdf& B = *H.bdf;
  B.st = Stats{};
  const bool chunks = p.solve_chunks != 0, has_save = p.save_interval >= 0, variable = n_stops > 0;
  int64_t n_chunks = 1;
  if (chunks) n_chunks = (int64_t)std::floor(p.tspan1 / p.solve_chunkstep + 0.5);
  std::vector<double> save_local;
  const double span_len = chunks ? p.solve_chunkstep : p.tspan1 - p.tspan0;
  if (chunks || has_save) {
    const double si = has_save ? p.save_interval : p.solve_chunkstep;
    const double base = chunks ? 0.0 : p.tspan0, last = chunks ? p.solve_chunkstep : p.tspan1;
    const int64_t cnt = (int64_t)std::floor(span_len / si + 1e-9) + 1;
    for (int64_t i = 0; i < cnt; i++) save_local.push_back(std::min(base + (double)i * si, last));
    if (!chunks && save_local.back() < last) save_local.push_back(last);
    if (chunks && std::fabs(save_local.back() - last) <= 1e-9 * last) save_local.back() = last;
  }
  const int64_t L = (int64_t)save_local.size();
  const bool save_hits_end = chunks && L > 0 && save_local.back() == p.solve_chunkstep;
  double abstol = p.abstol, reltol = p.reltol;
  B.set_tols(abstol, reltol);
  B.ban_negatives = p.ban_negatives != 0;
  B.lu_reuse = p.lu_reuse;
  B.step_thresh = p.step_thresh;
  B.lu_cache = (int)p.lu_cache;
  B.clear_cache(); B.n_cache_hits = 0; B.force_fresh_lu = false; B.steps_since_jac = 0;
  {
    const double xx = std::fabs(chunks ? p.solve_chunkstep : p.tspan1);
    B.dtmin = p.dtmin > 0.0 ? p.dtmin : std::nextafter(xx, INF) - xx;
  }
  B.k = k0;
  H.sol_t.clear(); H.sol_u.clear();
  auto push = [&](double tt, const double* u) { H.sol_t.push_back(tt); H.sol_u.insert(H.sol_u.end(), u, u + N); };
  std::vector<double> y(u0, u0 + N), y_start(N), row(N);
  int64_t next_stop = 0, rates_in_force = -1;
  int retcode = RET_SUCCESS;
  for (int64_t nc = 0; nc < n_chunks && retcode == RET_SUCCESS; nc++) {
    B.st.n_chunks++;
    const double t_start_g = chunks ? p.solve_chunkstep * (double)nc : p.tspan0;
    const double t_end_g = chunks ? t_start_g + p.solve_chunkstep : p.tspan1;
    const double shift = chunks ? (double)nc * p.solve_chunkstep : 0.0;
    const double t_loc0 = chunks ? 0.0 : p.tspan0, t_loc1 = chunks ? p.solve_chunkstep : p.tspan1;
    y_start = y;
    B.cache_suspended = false;
    const size_t n_out_start = H.sol_t.size();
    int attempts = 0;
    for (;;) {
      attempts++;
      retcode = RET_SUCCESS;
      B.iters_left = p.maxiters;
      int64_t stop_i = next_stop;
      while (variable && stop_i < n_stops && tstops[stop_i] <= t_start_g) stop_i++;
      if (variable) {
        const int64_t want = stop_i > 0 ? stop_i - 1 : 0;
        if (want != rates_in_force) { rates_in_force = want; B.k = k_table + (size_t)want * R; }
      }
      bool failed = false;
      int64_t save_i = 0;
      if (L > 0) { push(save_local[0] + shift, y.data()); save_i = 1; }
      else push(t_loc0 + shift, y.data());
      double t_seg = t_loc0;
      while (t_seg < t_loc1 && !failed) {
        double seg_end = t_loc1;
        bool ends_at_stop = false;
        if (variable && stop_i < n_stops && tstops[stop_i] < t_end_g) {
          const double loc = tstops[stop_i] - shift;
          if (loc < t_loc1) { seg_end = loc; ends_at_stop = true; }
        }
        if (seg_end > t_seg) {
          const double seg_len = seg_end - t_seg;
          if (!B.restart(0.0, y.data(), seg_len)) { retcode = RET_UNSTABLE; failed = true; break; }
          while (B.t < seg_len) {
            const Bdf::Status ss = B.step(seg_len);
            if (ss == Bdf::MAXITERS) { retcode = RET_MAXITERS; failed = true; break; }
            if (ss == Bdf::DTMIN) { retcode = RET_DTMIN; failed = true; break; }
            if (ss == Bdf::UNSTABLE) { retcode = RET_UNSTABLE; failed = true; break; }
            const double t_abs = B.t >= seg_len ? seg_end : t_seg + B.t;
            if (L > 0) {
              const int64_t last_i = (chunks && !(nc == n_chunks - 1 && !save_hits_end)) ? L - 1 : L;
              while (save_i < last_i && save_local[save_i] <= t_abs) {
                B.interpolate(std::min(save_local[save_i] - t_seg, B.t), row.data());
                push(save_local[save_i] + shift, row.data());
                save_i++;
              }
            } else {
              push(t_abs + shift, B.D.data());
            }
            B.select_order();
          }
          if (failed) break;
          std::copy(B.D.begin(), B.D.begin() + N, y.begin());
        }
        t_seg = seg_end;
        if (ends_at_stop) { B.k = k_table + (size_t)stop_i * R; rates_in_force = stop_i; stop_i++; }
      }
      if (!failed) {
        if (chunks && nc == n_chunks - 1 && L > 1 && save_hits_end) push(save_local[L - 1] + shift, y.data());
        next_stop = stop_i;
        break;
      }
      if (!p.adaptive_tols || attempts >= 5 || abstol / 10 <= EPS || reltol / 10 <= EPS) break;
      abstol /= 10; reltol /= 10;
      rates_in_force = -1;
      B.set_tols(abstol, reltol);
      B.st.n_retries++;
      B.clear_cache(); B.lu_valid = false; B.cache_suspended = true;
      y = y_start;
      for (double& v : y) if (v < 0.0) v = 0.0;      // the rescue path clips inherited negative concentrations (solver.cpp, solve_entry)
      H.sol_t.resize(n_out_start);
      H.sol_u.resize(n_out_start * (size_t)N);
    }
  }
  B.st.final_abstol = abstol; B.st.final_reltol = reltol;
  B.st.lu_nnz = B.lu.nnz(); B.st.lu_full = B.lu.n_full; B.st.lu_refactor = B.lu.n_refactor;
  B.st.wall_seconds = now_s() - wall0;
  H.st = B.st;
  return retcode;
}

}  // namespace

extern "C" {

void* cpub_create(int64_t N, int64_t R, const int64_t* rp, const int64_t* ri, const int64_t* rs, const int64_t* pp,
                  const int64_t* pi, const int64_t* ps) {
  Handle* H = new Handle();
  Net& n = H->net;
  n.N = N; n.R = R;
  n.rp.assign(rp, rp + R + 1); n.pp.assign(pp, pp + R + 1);
  n.ri.assign(ri, ri + rp[R]); n.rs.assign(rs, rs + rp[R]);
  n.pi.assign(pi, pi + pp[R]); n.ps.assign(ps, ps + pp[R]);
  n.build_jac_pattern();
  return H;
}
void cpub_destroy(void* h) { delete (Handle*)h; }

int cpub_solve(void* h, const CpuParams* p, const double* u0, const double* k0, const double* tstops, const double* k_table,
               int64_t n_stops, int64_t* n_saved, Stats* stats) {
  Handle& H = *(Handle*)h;
  const int rc = solve(H, *p, u0, k0, tstops, k_table, n_stops);
  if (n_saved) *n_saved = (int64_t)H.sol_t.size();
  if (stats) *stats = H.st;
  return rc;
}
void cpub_solution_copy(void* h, double* t, double* u) {
  Handle& H = *(Handle*)h;
  std::copy(H.sol_t.begin(), H.sol_t.end(), t);
  std::copy(H.sol_u.begin(), H.sol_u.end(), u);
}
void cpub_rhs(void* h, const double* k, const double* u, double* du) { ((Handle*)h)->net.rhs(k, u, du); }
int64_t cpub_jac_nnz(void* h) { return (int64_t)((Handle*)h)->net.jri.size(); }
void cpub_jac(void* h, const double* k, const double* u, int32_t* colptr, int32_t* rowidx, double* vals) {
  Net& n = ((Handle*)h)->net;
  if (colptr) std::copy(n.jcp.begin(), n.jcp.end(), colptr);
  if (rowidx) std::copy(n.jri.begin(), n.jri.end(), rowidx);
  if (vals) n.jac(k, u, vals);
}
// diagnostic: (I - c J(u)) x = b through the LU (first call: pivoting factorisation, later calls: refactorisation);
// returns nnz(L + U)
int64_t cpub_newton_solve(void* h, double c, const double* k, const double* u, const double* b, double* x) {
  Handle& H = *(Handle*)h;
  if (!H.bdf) H.bdf = new Bdf(H.net);
  Bdf& B = *H.bdf;
  B.k = k;
  B.eval_jac(u);
  if (!B.factor(c)) return -1;
  B.lu.solve(b, x, B.work);
  return B.lu.nnz();
}

}  // extern "C"
