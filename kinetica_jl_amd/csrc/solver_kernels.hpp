// Kernel launchers and device-visible structs of the BDF integrator (solver_kernels.hip).
#pragma once
#include <cmath>

#include "common.hpp"

namespace kin {

constexpr int BDF_MAX_ORDER = 5;
constexpr int BDF_NEWTON_MAXITER = 4;
constexpr int BDF_D_ROWS = BDF_MAX_ORDER + 3;

// Corrector tolerance as a fraction of the error weight atol + rtol |y| (the estimated iteration error must get below it):
// 0.03 at the default relative tolerance 1e-8 and looser ones, 0.1 (CVODE's nlscoef) from 1e-9 down, 1e-10 / rtol between.
// What it rests on (profiles/r05_newton_tol_ab.txt, one MI355X):
//  * at rtol <= 1e-9 an iteration asked to converge to 0.03 of the weight asks for less than the rounding of the right-hand
//    side's sums leaves (~1e-10 relative on these networks): it fails, and every failure restarts the step at a quarter. With 0.1
//    the 200-species solve at 1e-12 / 1e-10 ends 2-7 x closer to its Radau truth in half the steps (resident kernel rms 385 -> 55
//    tight units, host-driven 885 -> 394, CPU port 153 -> 61), C3 at 1e-11 / 1e-9 takes 0.60 s instead of 0.90 s;
//  * at 1e-8 a flat 0.1 is 7 % faster on C3 (0.350 -> 0.325 s) and passes every sweep with the default switches, but over the 140
//    solves of tools/robustness_sweep.py wide under four perturbed configurations (corrector fused, reuse band 0.3, 8 and 1
//    factorisation slots) 10 of 560 needed a tolerance retry after a step-size collapse, against 3 of 560 at 0.03 and 5 at 0.05
//    (ode15s) - all on 1 000-species networks, the ones that collapsed under 0.05 in round 2: not adopted there.
// Same rule: resident_core.hpp res_newton_frac, oracle/bdf.py, oracle/cpu_bdf.cpp.
// An ACCEPTED step (corrector converged, error test passed) that leaves a species below -BDF_NEG_DEEP error weights ends the
// segment as Unstable. The error test bounds what ONE step can do to one species at sqrt(N) / error constant weights in the worst
// case (~300 at 1 000 species; more only if a step's whole error sat on a single species of a larger network), so such a state
// has been growing over many accepted steps: it is the negative excursion of DESIGN 4 - below zero some species are unstable
// under mass-action kinetics, |u| grows with an e-folding time of ~0.1 ms and h follows it down for 500-1 400 more steps until
// dtmin or a non-finite state ends the attempt anyway. The chunk's tolerance retry (negative entries of its start state zeroed)
// carries the solve in either case; a false alarm costs one such retry (docs/DESIGN_HISTORY.md R5.12 has the measurements). The flag
// rides in the sum that counts negative entries: a thread contributes 1 for a negative entry, BDF_NEG_MARK for a deep one.
constexpr double BDF_NEG_DEEP = 1e3;
constexpr double BDF_NEG_MARK = 4294967296.0;   // 2^32 > any count of species
inline double bdf_newton_frac(double rtol) { return std::fmin(0.1, std::fmax(0.03, 1e-10 / rtol)); }

struct BdfCoef {  // passed to kernels by value
  double gamma[BDF_MAX_ORDER + 1];
  double alpha[BDF_MAX_ORDER + 1];
  double error_const[BDF_MAX_ORDER + 2];
};
struct BdfMat { double v[BDF_MAX_ORDER + 1][BDF_MAX_ORDER + 1]; };
struct BdfVec { double v[BDF_MAX_ORDER + 1]; };
struct RkVec { double v[7]; };   // per-stage weights of the explicit Dormand-Prince path

// device-resident control block, copied to the host once per step attempt
struct BdfCtrl {
  double dy_norm_old, dy_norm;
  double err_norm, err_m_norm, err_p_norm;
  double crate;        // contraction rate carried by the factorisation in use (CVODE's crate): set from the host's per-slot copy
                       // at iteration 0, updated by every later iteration, read back by the host with the step's result
  double scratch[4];   // bdf_norms_kernel: rms(y / w), rms(f0 / w), rms((f1 - f0) / w), max |f0| / (0.1 |y| + w)
  int newton_done, converged, n_iter, nonfinite, any_negative;
  int ticket;   // arrival counter of the multi-workgroup reductions (back to 0 when a launch ends)
  int lu_bad;   // set by a factorisation that met a vanishing pivot (|multiplier| > 1e8); cleared by the host, NOT by the predictor
  int spec_go;  // written by every corrector decision: 1 = the attempt ended as an accepted step (what the host will conclude
                // from the same numbers), so a speculatively enqueued next step may run (solver.cpp: speculation)
};

// LU cache: per-slot copies of diag(J) and the drift test of Solver::restart (at most LU_MAX_SLOTS slots)
constexpr int LU_MAX_SLOTS = 128;
struct SlotDriftArgs { const double* jd[LU_MAX_SLOTS]; double c[LU_MAX_SLOTS]; };
void launch_jac_diag(int N, const double* jv, const int32_t* j_diag, double* jd, hipStream_t s);
void launch_jac_diag_absmax(int N, const double* jv, const int32_t* j_diag, double* out, hipStream_t s);   // out[0] = max |J_ii|
void launch_slot_drift(int N, int n_slots, const double* jv, const int32_t* j_diag, const SlotDriftArgs& a, double* out, hipStream_t s);

void launch_bdf_predict(int N, int order, const double* D, const BdfCoef& cf, double atol, double rtol, double* y, double* psi,
                        double* d, double* scale, BdfCtrl* ctrl, hipStream_t s);   // also clears *ctrl
// The same with the solve's last gather stage in the same launch (solver_kernels.hip: stagec_newton_kernel). W = the slot's
// value array; the plan is SparseLU::stageC, whose aux entries are species indices.
struct NewtonFuse {
  const int* skip;
  int N, m;                     // species; size of the dense block
  int32_t off_x;                // W position of the dense block's solution
  const int32_t* x2_species;    // m: species behind the dense block's rows
  const double* scale; double* y; double* d; const double* D;
  int order;
  double upd, atol, rtol, ec, ec_m, ec_p;   // error constants of order, order - 1, order + 1
  int iter, maxit;
  double tol, rate_max, crate0, tol_first, dy_first_max;
  int crate_from_ctrl, ban_negatives;   // see launch_bdf_newton
  BdfCtrl* ctrl; double* part; BdfCtrl* host_ctrl; unsigned long long* host_seq; unsigned long long seq; int publish_always;
};
struct SegPlanView;
int stagec_newton_grid(const SegPlanView& p, int m);   // workgroups of the launch (5 partial sums each in `part`)
void launch_stagec_newton(const SegPlanView& p, double* W, const NewtonFuse& f, hipStream_t s);

// One corrector iteration's update + decision + (folded in) the step's error estimate; the launch that decides publishes
// the control block to host_ctrl / host_seq (device-visible pinned host memory, or null), the batch's last launch
// (`publish_always`) also when nothing is decided yet. dy = upd * x; crate0: carried rate, whose first-iteration test needs
// the estimated remaining error below tol_first (< 0: test off) and dy_norm <= dy_first_max.
void launch_bdf_newton(int N, int iter, int maxit, double tol, const int32_t* xloc, const double* W, const double* scale,
                       double* y, double* d, double upd, double rate_max, double crate0, double tol_first, double dy_first_max,
                       int order, const double* D, double atol, double rtol, const BdfCoef& cf, BdfCtrl* ctrl, double* part,
                       BdfCtrl* host_ctrl, unsigned long long* host_seq, unsigned long long seq, bool publish_always, hipStream_t s,
                       bool crate_from_ctrl = false, bool ban_negatives = false);
// crate_from_ctrl: iteration 0 takes the carried rate from ctrl->crate (what the previous step left there) instead of
// crate0 - a speculatively enqueued step cannot know it on the host; ban_negatives: part of the `spec_go` verdict.
// `part`: bdf_reduce_slot() * bdf_reduce_blocks(N) doubles of partial sums
int bdf_reduce_slot();   // doubles per workgroup in the `part` buffer of the corrector launches
int bdf_reduce_blocks(int N);
void launch_bdf_accept(int N, int order, double* D, const double* d, double* copy_out, hipStream_t s);   // copy_out (optional): the new state
// accept of the previous step (order `ao`) + predictor of the next one in one pass (the host defers the accept)
void launch_bdf_accept_predict(int N, int ao, int order, double* D, const BdfCoef& cf, double atol, double rtol, double* y, double* psi,
                               double* d, double* scale, BdfCtrl* ctrl, double* copy_out, hipStream_t s, const int* go = nullptr);
// `go` (optional device flag): the launch does nothing but end the batch behind it (newton_done = 1) when *go == 0
void launch_bdf_change_D(int N, int order, const BdfMat& ru, double* D, hipStream_t s);
void launch_bdf_init_D(int N, int nrows, const double* y0, const double* f0, double h, double* D, hipStream_t s);
void launch_bdf_interp(int N, int order, const double* D, const BdfVec& p, double* out, hipStream_t s);
void launch_rk_combine(int N, int n, const RkVec& w, const double* y, const double* K, double* out, hipStream_t s);
void launch_rk_error(int N, const RkVec& e, const double* y, const double* y_new, const double* K, double atol, double rtol,
                     BdfCtrl* ctrl, double* part, BdfCtrl* host_ctrl, unsigned long long* host_seq, unsigned long long seq,
                     hipStream_t s);
void launch_axpy_out(int N, const double* a, double sc, const double* b, double* out, hipStream_t s);
void launch_clip_negative(int N, const double* a, double* out, hipStream_t s);   // out = max(a, 0)
void launch_bdf_norms(int N, const double* y0, const double* f0, const double* f1, double atol, double rtol, BdfCtrl* ctrl, hipStream_t s);
void launch_rowdot(int N, int64_t M, const double* U, const double* w, double* out, hipStream_t s);
void launch_colmax(int N, int64_t M, const double* U, double* out, hipStream_t s);
struct ArrheniusAt;
void launch_rates_skip_T(int64_t R, const ArrheniusAt& at, double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate,
                         const int* skip, hipStream_t s);
void launch_rates_skip(int64_t R, const double* k, const double* u, const int32_t* x0, const int32_t* x1, double* rate,
                       const int* skip, hipStream_t s);

}  // namespace kin
