// Host-side preparation shared by the resident integrator's owner (resident.cpp) and its CPU replay (tests/native/):
// the save grid, chunk count and dtmin of a solve exactly as solve_entry derives them (solver.cpp; reference
// src/solving/methods.jl:756-758, 829-846, 164, 232, 694, 770), and the integrator settings with their environment switches.
#pragma once
#include <cmath>
#include <cstdlib>
#include <limits>
#include <vector>

#include "../../include/kinetica_hip.h"
#include "resident_core.hpp"

namespace kin {

struct ResGrid {
  std::vector<double> save_local;
  int64_t n_chunks = 1, cap = 0;
  bool save_hits_end = false;
  double dtmin = 0.0;
};

// requires a save grid (chunkwise, or a save_interval): the resident path writes into a solution buffer of known size
inline bool res_has_grid(const kin_params& p) { return p.solve_chunks != 0 || p.save_interval >= 0; }

inline ResGrid make_res_grid(const kin_params& p) {
  ResGrid g;
  const double INF = std::numeric_limits<double>::infinity();
  const bool chunks = p.solve_chunks != 0, has_save = p.save_interval >= 0;
  if (chunks) g.n_chunks = (int64_t)(p.tspan1 / p.solve_chunkstep);
  const double span_len = chunks ? p.solve_chunkstep : (p.tspan1 - p.tspan0);
  const double si = has_save ? p.save_interval : p.solve_chunkstep;
  const double base = chunks ? 0.0 : p.tspan0;
  const double last = chunks ? p.solve_chunkstep : p.tspan1;
  const int64_t cnt = (int64_t)std::floor(span_len / si + 1e-9) + 1;
  for (int64_t i = 0; i < cnt; i++) g.save_local.push_back(std::min(base + (double)i * si, last));
  if (!chunks && g.save_local.back() < last) g.save_local.push_back(last);
  if (chunks && std::fabs(g.save_local.back() - last) <= 1e-9 * last) g.save_local.back() = last;
  const int64_t L = (int64_t)g.save_local.size();
  g.save_hits_end = chunks && L > 0 && g.save_local.back() == p.solve_chunkstep;
  g.cap = chunks ? (L - 1) * g.n_chunks + 1 : L;
  if (p.dtmin > 0.0) g.dtmin = p.dtmin;
  else {
    const double x = std::fabs(chunks ? p.solve_chunkstep : p.tspan1);
    g.dtmin = std::nextafter(x, INF) - x;
  }
  return g;
}

// integrator settings: the constants of solver.cpp (Solver) and its two cache switches (KIN_LU_CACHE_SLOTS, KIN_LU_BAND)
inline void res_default_settings(ResParams& P, int n_slots_max) {
  auto envd = [](const char* n, double d) { const char* e = getenv(n); return e ? atof(e) : d; };
  auto envi = [](const char* n, long long d) { const char* e = getenv(n); return e ? atoll(e) : d; };
  int want = (int)envi("KIN_LU_CACHE_SLOTS", RES_MAX_SLOTS);
  want = std::max(1, std::min(want, std::min(n_slots_max, RES_MAX_SLOTS)));
  const double band = envd("KIN_LU_BAND", 0.35);
  P.n_slots = want;
  P.lu_band = want > 1 ? band : (getenv("KIN_LU_BAND") ? band : 0.0);
  // the constants of solver.cpp (Solver): reuse_rate_max, crate_dy_max, lu_drift_max, corrector tolerance, crate_max_age, lu_max_age
  P.reuse_rate_max = 0.15;
  P.crate_dy_max = 0.2;
  P.lu_drift_max = 1.0;
  P.newton_frac = -1.0;   // < 0: bdf_newton_frac(rtol), the rule of solver_kernels.hpp (a test may pin a value here)
  P.crate_max_age = 10;
  P.lu_max_age = 50;
  P.carry_rate = 1;
}

inline void res_fill_params(ResParams& P, const kin_params& p, const ResGrid& g) {
  P.tspan0 = p.tspan0; P.tspan1 = p.tspan1; P.abstol = p.abstol; P.reltol = p.reltol; P.chunkstep = p.solve_chunkstep;
  P.dtmin = g.dtmin;
  P.solve_chunks = p.solve_chunks == 2 ? 2 : (p.solve_chunks != 0 ? 1 : 0); P.adaptive_tols = p.adaptive_tols != 0; P.ban_negatives = p.ban_negatives != 0;
  P.save_hits_end = g.save_hits_end ? 1 : 0;
  P.maxiters = p.maxiters; P.n_chunks = g.n_chunks;
  P.L = (int32_t)g.save_local.size();
  P.sol_cap = g.cap;
}

}  // namespace kin
