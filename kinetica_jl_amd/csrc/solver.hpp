// BDF integrator + chunk/tstop driver behind kin_solve (solver.cpp, lu.cpp, solver_kernels.hip).
#pragma once
#include <string>
#include "../../include/kinetica_hip.h"
#include "handle.hpp"

namespace kin {

// Runs the whole solve (chunk loop, discrete rate updates, retry loop); stores the solution in
// the handle; returns the final KIN_RETCODE_*.
int solve_entry(kin_network* h, const kin_params& p, const double* u0, const double* tstops,
                const double* T_stops, const double* k_table, int64_t n_stops, kin_stats* stats,
                const double* t_nodes = nullptr, const double* T_nodes = nullptr, int64_t n_nodes = 0,
                bool explicit_solver = false);   // explicit: Dormand-Prince 5(4) instead of the BDF (kin_solve_explicit)
// return_integrator=true: initialise / advance / inspect the integrator without solving (solver.cpp)
void integrator_init(kin_network* h, const kin_params& p, const double* u0, const double* tstops, const double* T_stops,
                     const double* k_table, int64_t n_stops, const double* t_nodes = nullptr, const double* T_nodes = nullptr,
                     int64_t n_nodes = 0);   // n_nodes > 0: continuous rate updates (kin_integrator_init_continuous)
int64_t integrator_step(kin_network* h, int64_t max_steps);
void integrator_state(kin_network* h, double* t, double* u, int32_t* retcode, kin_stats* stats);
// Resident integrator (resident.cpp): the whole solve in one launch, one workgroup per trajectory. `resident_eligible`:
// the network is small enough (KIN_RESIDENT_MAX_N, default 400 species: measured crossover against the host-driven path), the call has a save grid and uses none of the
// features that stay on the host-driven path (continuous rates, explicit solver, manual stepping, traces, fault injection);
// KIN_RESIDENT=0 switches the path off.
bool resident_eligible(kin_network* h, const kin_params& p, bool continuous, bool explicit_solver);
int resident_solve(kin_network* h, const kin_params& p, const double* u0, const double* tstops, const double* T_stops,
                   const double* k_table, int64_t n_stops, kin_stats* stats);
void resident_ensemble(kin_network* h, const kin_params& p, int64_t K, const double* u0, const double* k, const double* T,
                       const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* out_rows,
                       double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats);
// does the network fit the resident kernel (its state, rates and solve vectors in one compute unit's LDS)?
bool resident_fits(kin_network* h);
// ... and does an ensemble of K members take the one-launch form (resident.cpp: not few members of a network at the kernel's upper end)?
bool resident_ensemble_route(kin_network* h, int64_t K);
// can the lockstep form (ensemble.cpp) take this network's factorisation? (fused solve with a dense Schur block)
bool ensemble_batched_supported(kin_network* h, std::string* why);
// K members of a network beyond that, advanced in lockstep rounds of batched launches (ensemble.cpp); same arguments
void batched_ensemble(kin_network* h, const kin_params& p, int64_t K, const double* u0, const double* k, const double* T,
                      const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* out_rows,
                      double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats);
// validation of a solve's arguments (ODESimulationParams constructor, params.jl:77-104, and the rate inputs); throws
void validate_solve(kin_network* h, const kin_params& p, const double* tstops, const double* T_stops, const double* k_table,
                    int64_t n_stops, const double* t_nodes, const double* T_nodes, int64_t n_nodes, bool need_handle_rates);
// max over saved times per species, reduced on the device
void solution_max(kin_network* h, double* out_umax);
// diagnostic: (I - c J(u)) x = b through the solver's LU
void newton_solve(kin_network* h, double c, const double* u, const double* b, double* x);

}  // namespace kin
