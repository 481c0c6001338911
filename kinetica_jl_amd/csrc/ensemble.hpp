// Batched ensemble of LARGE networks (ensemble.cpp, ensemble_kernels.inc): K trajectories of one network advance in
// lockstep rounds; every launch of a round carries all members that need that kind of work (blockIdx.y = entry of the
// round's list). The controller of every member is resident_core.hpp - the one the resident kernel and its CPU replay
// run - on a host thread of its own; its backend hands each operation to the round instead of launching it.
// Why: K handles on K host threads (round 3) top out at ~17 solves/s on the 10k-species network whatever K is - every
// member's step is a chain of ~14 small dependent dispatches and the chip takes ~200 k dispatches/s over all its queues
// (DESIGN 7); a round of this path is ONE such chain for all K members.
// Networks that fit one compute unit's LDS take the resident kernel instead (resident.cpp), one workgroup per member.
#pragma once
#include "kernels.hpp"
#include "resident_core.hpp"
#include "solver_kernels.hpp"

namespace kin {

// per-member state (device pointers; constant for the lifetime of an ensemble solve)
struct EnsRep {
  double *D, *y, *psi, *d, *scale, *f0, *f1, *ytmp, *cs, *jv, *rate, *dr, *k, *part;
  BdfCtrl* ctrl;
};

// one entry of a round's list
struct EnsOp {
  int32_t rep, i0, i1, i2;       // member; operation-specific integers
  double d0, d1;                 // operation-specific scalars
  double* W;                     // corrector: value array of the factorisation in use
  const double* sinv;            // ... and where its dense inverse sits
  double* out;                   // save / dense output: destination row; load_u0: source
  ResCorrIn in;                  // corrector
  double ru[36];                 // change_D: (R U), 6 x 6, identity outside the orders involved
  double p[8];                   // dense-output weights
};

enum EnsVecOp : int { EV_LOAD_U0 = 0, EV_CS_FROM_Y, EV_Y_FROM_CS_CLIPPED, EV_Y_FROM_D0, EV_YTMP_FROM_D0, EV_YTMP_AXPY, EV_SAVE_Y, EV_INTERP };

// launchers: `n` entries in d_ops (device copy of the round's list), reps = the ensemble's member table
void ens_vec(int N, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);
void ens_accept(int N, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);          // i0 = order
void ens_change_D(int N, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);        // ru
void ens_init_D(int N, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);          // i0 = from ytmp, d0 = h
void ens_norms(int N, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);           // i0 = with f1, d0 / d1 = atol / rtol
// right-hand side: i0 = 0: y -> f0, 1: ytmp -> f1, 2: ytmp -> f0
void ens_rhs(int N, int R, const int32_t* x0, const int32_t* x1, const SegPlanView& rhs_plan, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);
void ens_jac(int R, const int32_t* x0, const int32_t* x1, const SegPlanView& jac_plan, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);
// rate constants: i0 = 1: copy of the row `out`; 2: Arrhenius at temperature d0
void ens_apply_rates(int R, const double* Ea, const double* A, int has_kmax, double k_max, double t_mult, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);
// a corrector attempt of every entry: predictor + `iters` iterations (each: rates, residual, the fused solve's three stages,
// update + decision; launches behind a member's decision are no-ops for it)
struct EnsSolveTables {
  int N, R, m, mpad, ns;
  int64_t off_y, off_x;
  const int32_t *x0, *x1, *xloc, *x2_species;
  SegPlanView resid, stageA, stageC;
  BdfCoef cf;
};
void ens_predict(const EnsSolveTables& T, const EnsRep* reps, const EnsOp* d_ops, int n, hipStream_t s);
void ens_iterations(const EnsSolveTables& T, const EnsRep* reps, const EnsOp* d_ops, int n, int it0, int iters, hipStream_t s);
int ens_reduce_doubles(int N);   // doubles of EnsRep::part
// the members' control blocks into pinned host memory + a sequence number behind them (a round's hand-over without a stream sync)
void ens_publish(const BdfCtrl* ctrl, BdfCtrl* host_ctrl_dev, int K, unsigned long long* host_seq_dev, unsigned long long seq, hipStream_t s);

}  // namespace kin
