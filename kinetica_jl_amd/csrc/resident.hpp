// Resident integrator: device-visible tables and the launcher (resident.hip), host-side owner (resident.cpp).
// One 512-thread workgroup (8 wavefronts, 256 VGPRs each) integrates one trajectory from u0 to the end of the time span - chunk loop, rate updates,
// retries, BDF steps, Jacobians, factorisations and corrector iterations - without leaving the GPU (resident_core.hpp).
#pragma once
#include "kernels.hpp"
#include "resident_core.hpp"

namespace kin {

constexpr int RES_MAX_ROUNDS = 32;

// solve modes of the factorised Newton matrix (lu.hpp)
enum : int { RES_SOLVE_FUSED = 0, RES_SOLVE_EXPLICIT = 1, RES_SOLVE_PLAIN = 2 };

// immutable per network: reaction tables, gather plans, symbolic LU (device pointers)
struct ResNetDev {
  int32_t N, R, nnzJ, ns, m, mpad, nrounds, n_mono_ent, solve_mode, has_kmax;
  int32_t desc_in_lds, pad0_;      // the launch provisions LDS for the task descriptors of resid_plan / stageA / stageC
  int64_t off_diag, off_U, off_L, off_S, off_y, off_x, off_dinv, off_vec_end, w_size;
  double k_max, t_mult;
  const int32_t *x0, *x1, *jmap, *ent_pivot, *yloc, *xloc, *j_diag;
  const int32_t *mono_ent_ptr, *mono_ptr, *mono_fac, *mono_dst;
  const float* mono_sign;
  const double *Ea, *A;            // rate_mode 2
  const double *k_table, *T_stops; // rate_mode 1 / 2
  int32_t round_e0[RES_MAX_ROUNDS + 1];   // first L entry of every elimination round
  SegPlanView rhs_plan, jac_plan, resid_plan, lz_build, nvu_build, stageA, stageC, fwdZ, fwd_dense, bwdT, bwdV;
  SegPlanView schur[RES_MAX_ROUNDS], fwd[RES_MAX_ROUNDS], bwd[RES_MAX_ROUNDS];
};

// per trajectory: inputs, work vectors, LU-cache slots, outputs
struct ResTrajDev {
  const double* u0;
  double *k, *D, *y, *psi, *d, *scale, *f0, *f1, *ytmp, *chunk_start, *jv, *rate, *dr;
  double* gj_scratch;   // mpad x mpad: second block of the dense inverse's ping-pong
  double* W;       // n_slots value arrays of w_size doubles
  double* jd;      // n_slots copies of diag(J) (drift guard of the LU cache)
  double *sol, *sol_t;
  ResResult* result;
};

// enqueues the solve of K trajectories (grid = K workgroups of RES_WG = 512 threads)
// `m`: dimension of the dense Schur block (sizes the dynamic LDS of its row panel; at most RES_MAX_DENSE)
constexpr int RES_MAX_DENSE = 512;
// dynamic LDS of the kernel: y, d, psi, scale (4 N), the solve-vector window of W, max(R, row panel of the dense inverse)
size_t resident_dyn_lds(int N, int R, int m, int64_t window);
size_t resident_desc_bytes(const SegPlanView& resid, const SegPlanView& stageA, const SegPlanView& stageC);   // ... + these, when they fit
constexpr size_t RES_LDS_BUDGET = 160 * 1024 - 16 * 1024;   // what is left of a CU's LDS next to the kernel's static blocks
void launch_resident(int K, size_t dyn_lds, const ResNetDev* d_net, const ResTrajDev* d_traj, const ResParams* d_par, hipStream_t s);
// the same kernel built with half the registers per lane: two workgroups share a compute unit (resident_w4.hip)
void launch_resident_shared_cu(int K, size_t dyn_lds, const ResNetDev* d_net, const ResTrajDev* d_traj, const ResParams* d_par, hipStream_t s);
size_t resident_static_lds();    // static LDS of the kernel (what two co-resident workgroups need twice, next to 2 x dyn_lds)

}  // namespace kin
