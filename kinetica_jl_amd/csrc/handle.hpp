// The object behind the opaque kin_network handle of include/kinetica_hip.h.
#pragma once
#include <memory>
#include <string>

#include "common.hpp"
#include "kernels.hpp"
#include "network.hpp"
#include "tiled.hpp"

namespace kin {
struct Solver; struct IntegratorState; struct ResidentSolver; struct EnsembleSolver;
struct ResidentDeleter { void operator()(ResidentSolver* p) const; };   // resident.cpp (the type is complete there only)
struct EnsembleDeleter { void operator()(EnsembleSolver* p) const; };   // ensemble.cpp
}

struct kin_network {
  kin::NetworkHost host;
  hipStream_t stream = nullptr;
  std::string err;

  // reaction tables
  kin::DevBuf<int32_t> x0, x1, sp_ptr, sp_rxn;
  kin::DevBuf<uint32_t> sweep_rec;   // 4 words per reversible pair (kernels.hip: SweepRec)
  kin::DevBuf<int32_t> sweep_k;      // (kf, kr) per pair
  kin::DevBuf<uint32_t> sweep_rec64; // 64-bit packed records (register-resident sweep)
  kin::DevBuf<uint32_t> gen_rec8;    // 8-byte fixed-role records of the general LDS sweep
  kin::DevBuf<int32_t> gen_expl;     // its explicit-operand records (slow path)
  kin::DevBuf<int32_t> sweep_copy;   // species of the split hubs' extra accumulator entries
  kin::DevBuf<uint32_t> big_rec, big_rec8;   // large-N sweep: label-space records (16-byte slow-path and 8-byte stream format)
  kin::DevBuf<int32_t> big_spec, big_tptr, big_expl;
  kin::DevBuf<uint32_t> big_tent;    // (record, local label | coef << 24) pairs per tail tile
  kin::DevBuf<double> big_scratch;   // per-workgroup rows (tail u + net rates) of the large-N sweep (grown on demand)
  kin::DevBuf<float> sp_coef;
  kin::SegPlanDev rhs_plan, jac_plan;

  // rate constants / calculator
  kin::DevBuf<double> k, Ea, A, table, T_stops;
  bool has_rates = false, has_arrhenius = false, has_kmax = false;
  // Continuous-rate solves: a temperature whose rate constants have not been formed yet. The first kernel that reads k
  // (rates / operand derivatives / the corrector's rates) evaluates the Arrhenius law itself and stores k for the readers
  // behind it: no launch of its own per step attempt (the reference inlines k(T(t)) into the ODEs, methods.jl:389-419).
  bool k_pending = false;
  double T_pending = 0.0;
  void set_pending_T(double T) { T_pending = T; k_pending = true; has_rates = true; }
  kin::ArrheniusAt pending_at() const { return kin::ArrheniusAt{Ea.p, A.p, has_kmax ? 1 : 0, k_max, t_mult, T_pending}; }
  void flush_pending_T(hipStream_t s);   // forms k now if a temperature is still pending
  double k_max = 0.0, t_mult = 1.0;
  int64_t table_rows = 0;

  // single-state work vectors
  kin::DevBuf<double> u, du, rate, dr, jvals;

  // batched sweep workspace
  kin::DevBuf<double> b_u, b_k, b_du;

  // tiled sweep in library order (tiled.hpp, tiled_api.cpp): built at the first call that needs it
  kin::TiledHost tiled;
  bool tiled_tried = false;
  kin::DevBuf<uint32_t> t_rec;
  kin::DevBuf<int32_t> t_copy, t_kf, t_kr, t_rxn_of_slot, t_spec_of_lib, t_lib_of_spec, t_kslot, t_stage_lib, t_stage_off;
  kin::DevBuf<double> t_par, t_T, t_u, t_du;   // Arrhenius parameters per record (4 doubles), temperatures, layout scratch
  bool t_par_valid = false;

  // the device this handle lives on (the one current at kin_network_create) and its compute units
  int device = 0, n_cu = 0;

  // solver + stored solution (solver.cpp)
  std::unique_ptr<kin::Solver> solver;
  std::unique_ptr<kin::IntegratorState> integ;   // return_integrator=true stepping state
  std::unique_ptr<kin::ResidentSolver, kin::ResidentDeleter> resident;   // one-workgroup-per-trajectory integrator (resident.cpp)
  std::unique_ptr<kin::EnsembleSolver, kin::EnsembleDeleter> ensemble;   // lockstep ensemble of large networks (ensemble.cpp)
  // solve-only copies of this handle (own stream, work vectors, Solver and LU cache; no sweep tables): a SMALL ensemble of a
  // large network is K independent kin_solve calls on K host threads (capi.cpp: replica_ensemble)
  std::vector<kin_network*> replicas;
  size_t lu_budget_mb = 0;   // device memory of this handle's LU cache (0: KIN_LU_CACHE_MB, default 32768); set on replicas
  std::vector<double> sol_t, sol_u;
  kin::DevBuf<double> d_sol_u;   // saved states on the device, [n_saved][N]
  int64_t n_saved = 0;

  kin_network();
  ~kin_network();
  void rhs_dev(const double* d_u, double* d_du);        // du = f(u) with current k
  void jac_dev(const double* d_u, double* d_vals);      // CSR values with current k
  void sweep_dev(int64_t B, const double* d_u, const double* d_k, double* d_du, hipStream_t s);   // batched RHS
};
