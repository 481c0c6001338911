/*
 * kin_oracle.c - CPU restatement of the arithmetic on Kinetica.jl's solve path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under kinetica_jl_amd/ may link, import or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * as the checker / the timed CPU baseline, never as the product.
 *
 * PARITY UNPINNED for the solve itself: the reference's own tests hold no RHS, Jacobian,
 * rate-constant or trajectory vectors, the path is Julia (no toolchain in
 * the build image) and its numerics live in un-vendored packages (Catalyst 14.4,
 * ModelingToolkit 9, OrdinaryDiffEq 6.95, Sundials/KLU; Project.toml:36-60). What IS pinned
 * by reference material is checked in tests/test_oracle_golden.py: the Arrhenius functor
 * against the reference's arrhenius_params.bson data, the mass-action law against the
 * 5-species system written out in docs/src/tutorials/ode-solution.md:33-41, condition
 * profiles against test/Main/conditions.jl.
 *
 * Plain scalar C, one thread, summation in reaction order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* u^s for the small integer stoichiometries of a CRN (s >= 0). */
static double pow_int(double u, int64_t s) {
  double p = 1.0;
  for (int64_t i = 0; i < s; i++) p *= u;
  return p;
}

/*
 * Mass-action right-hand side, following make_rs (src/solving/solve_utils.jl:318-334):
 * Reaction(k[i], reactants, products, sr, sp) in a ReactionSystem with
 * combinatoric_ratelaws=false gives
 *     rate_r = k_r * prod_j u[id_reacs[r][j]]^stoic_reacs[r][j]        (no 1/s! factor)
 *     du_i  += (stoic_prods[i in r] - stoic_reacs[i in r]) * rate_r
 * (worked example: docs/src/tutorials/ode-solution.md:33-41). Indices are 0-based here.
 */
void orc_rhs(int64_t n, int64_t nr,
             const int64_t* reac_ptr, const int64_t* reac_idx, const int64_t* reac_sto,
             const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
             const double* k, const double* u, double* du) {
  memset(du, 0, (size_t)n * sizeof(double));
  for (int64_t r = 0; r < nr; r++) {
    double rate = k[r];
    for (int64_t p = reac_ptr[r]; p < reac_ptr[r + 1]; p++) rate *= pow_int(u[reac_idx[p]], reac_sto[p]);
    for (int64_t p = reac_ptr[r]; p < reac_ptr[r + 1]; p++) du[reac_idx[p]] -= (double)reac_sto[p] * rate;
    for (int64_t p = prod_ptr[r]; p < prod_ptr[r + 1]; p++) du[prod_idx[p]] += (double)prod_sto[p] * rate;
  }
}

/* B independent states at once (state-major u[b][n], k[b][nr] or one shared k when k_stride == 0):
 * the ensemble form of the same evaluation, one state per OpenMP thread. Only bench.py's all-core
 * CPU baseline uses it (the reference's solve path itself is single-threaded). */
void orc_rhs_many(int64_t n, int64_t nr,
                  const int64_t* reac_ptr, const int64_t* reac_idx, const int64_t* reac_sto,
                  const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
                  int64_t B, const double* k, int64_t k_stride, const double* u, double* du, int n_threads) {
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int64_t b = 0; b < B; b++)
    orc_rhs(n, nr, reac_ptr, reac_idx, reac_sto, prod_ptr, prod_idx, prod_sto, k + b * k_stride, u + b * n, du + b * n);
}

/* Per-reaction rates only (used by tests of the rate kernel). */
void orc_rates(int64_t nr, const int64_t* reac_ptr, const int64_t* reac_idx, const int64_t* reac_sto,
               const double* k, const double* u, double* rate) {
  for (int64_t r = 0; r < nr; r++) {
    double x = k[r];
    for (int64_t p = reac_ptr[r]; p < reac_ptr[r + 1]; p++) x *= pow_int(u[reac_idx[p]], reac_sto[p]);
    rate[r] = x;
  }
}

/*
 * Analytic Jacobian J[i][j] = d(du_i)/d(u_j), what ODEProblem(...; jac=true, sparse=true)
 * (src/solving/methods.jl:157-158) derives symbolically from the system above:
 *     J[i][j] += nu[i][r] * k_r * s_jr * u_j^(s_jr - 1) * prod_{l != j} u_l^(s_lr)
 * Emitted as COO triplets (duplicates allowed; the caller sums them). Returns the number
 * of triplets written; `cap` is the capacity of the three arrays (8 per reaction with
 * molecularity <= 2 per side is always enough: <=2 reactant columns x <=4 rows).
 */
int64_t orc_jac_coo(int64_t n, int64_t nr,
                    const int64_t* reac_ptr, const int64_t* reac_idx, const int64_t* reac_sto,
                    const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
                    const double* k, const double* u,
                    int64_t cap, int64_t* row, int64_t* col, double* val) {
  (void)n;
  int64_t m = 0;
  for (int64_t r = 0; r < nr; r++) {
    for (int64_t pj = reac_ptr[r]; pj < reac_ptr[r + 1]; pj++) {
      const int64_t j = reac_idx[pj];
      const int64_t sj = reac_sto[pj];
      double d = k[r] * (double)sj * pow_int(u[j], sj - 1);
      for (int64_t pl = reac_ptr[r]; pl < reac_ptr[r + 1]; pl++)
        if (pl != pj) d *= pow_int(u[reac_idx[pl]], reac_sto[pl]);
      for (int64_t p = reac_ptr[r]; p < reac_ptr[r + 1]; p++) {
        if (m >= cap) return -1;
        row[m] = reac_idx[p]; col[m] = j; val[m] = -(double)reac_sto[p] * d; m++;
      }
      for (int64_t p = prod_ptr[r]; p < prod_ptr[r + 1]; p++) {
        if (m >= cap) return -1;
        row[m] = prod_idx[p]; col[m] = j; val[m] = (double)prod_sto[p] * d; m++;
      }
    }
  }
  return m;
}

/*
 * PrecalculatedArrheniusCalculator functor (src/solving/calculator.jl:223-232):
 *     k_r = A .* exp.(-Ea / (R*T)) * N_A * t_mult
 *     with k_max:  1.0 ./ ((1.0 / k_max) .+ (1.0 ./ k_r))
 * Constants: src/constants.jl:4-5. Operation order follows the Julia expression
 * (left-to-right: ((A*exp(..))*N_A)*t_mult; -Ea/(R*T) with the product R*T formed first).
 * has_kmax == 0 is the `k_max = nothing` dispatch.
 */
#define ORC_R 8.314462618
#define ORC_NA 6.02214076e23
void orc_arrhenius(int64_t n, const double* Ea, const double* A, int has_kmax, double k_max,
                   double t_mult, double T, double* k) {
  const double RT = ORC_R * T;
  for (int64_t i = 0; i < n; i++) {
    double kr = A[i] * exp(-Ea[i] / RT) * ORC_NA * t_mult;
    k[i] = has_kmax ? 1.0 / ((1.0 / k_max) + (1.0 / kr)) : kr;
  }
}

/*
 * DummyKineticCalculator functor (src/solving/calculator.jl:130-132, 144-146): note the
 * time multiplier is applied AFTER the cap here (and before it in Arrhenius).
 */
void orc_dummy(int64_t n, const double* rates, int has_kmax, double k_max, double t_mult, double* k) {
  for (int64_t i = 0; i < n; i++)
    k[i] = has_kmax ? 1.0 / ((1.0 / k_max) + (1.0 / rates[i])) * t_mult : rates[i] * t_mult;
}

/*
 * Rate table of calculate_discrete_rates (src/solving/solve_utils.jl:91-109):
 * one Arrhenius evaluation per time stop, table[s][r].
 */
void orc_rate_table(int64_t n, const double* Ea, const double* A, int has_kmax, double k_max,
                    double t_mult, const double* T, int64_t n_stops, double* table) {
  for (int64_t s = 0; s < n_stops; s++) orc_arrhenius(n, Ea, A, has_kmax, k_max, t_mult, T[s], table + s * n);
}
