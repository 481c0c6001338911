/*
 * kinetica_hip.h - C ABI of libkinetica_hip.so, the MI355X (gfx950) implementation of
 * Kinetica.jl's kinetic-ODE solve path (src/solving in the reference).
 *
 * The reference has no FFI on this path (pure Julia multiple dispatch), so this
 * header DEFINES the boundary a `ccall` shim binds. Every entry
 * point cites the reference code it stands in for (paths relative to the reference
 * repository root). Conventions:
 *   - all entry points are extern "C", return an int status (KIN_OK == 0) and never throw;
 *   - plain pointers and sizes only; the caller owns every host buffer, the library
 *     never keeps a host pointer after a call returns;
 *   - Float64 / Int64 everywhere, as the reference (init_network(fType=Float64,
 *     iType=Int64), src/exploration/network.jl:491);
 *   - `index_base` is 1 when called from Julia (1-based species ids,
 *     src/exploration/network.jl:55-56) and 0 from C / Python;
 *   - one host thread per handle; handles are independent (one per GPU / stream);
 *   - a handle owns device memory; kin_network_destroy(NULL) is a no-op.
 * There is no CPU fallback: without a HIP device every compute call returns
 * KIN_ERR_DEVICE.
 */
#ifndef KINETICA_HIP_H
#define KINETICA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (error convention of SURVEY 8(b)) ------------------------------ */
enum {
  KIN_OK = 0,
  KIN_ERR_INVALID_ARG = 1,   /* ArgumentError in the reference (params.jl:77-104, calculator.jl:200-204) */
  KIN_ERR_UNSUPPORTED = 2,   /* molecularity > 2 on one side (network.jl:275-279) */
  KIN_ERR_DEVICE = 3,        /* HIP runtime error / no device */
  KIN_ERR_SOLVE_FAILED = 4,  /* ErrorException("ODE solution failed.") (solve_utils.jl:405-411) */
  KIN_ERR_CAPACITY = 5,      /* caller-provided output buffer too small */
  KIN_ERR_STATE = 6          /* call order violated (e.g. rates never set) */
};

/* ---- integrator return codes, 1:1 with SciMLBase.ReturnCode as consumed by
 *      successful_retcode (solve_utils.jl:391) and stored in the solution (methods.jl:862) */
enum {
  KIN_RETCODE_SUCCESS = 0,
  KIN_RETCODE_MAXITERS = 1,
  KIN_RETCODE_DTLESSTHANMIN = 2,
  KIN_RETCODE_UNSTABLE = 3     /* a non-finite state at (re)initialisation, or an accepted step that leaves a species below -1e3
                                * error weights (the blow-up of a negative concentration, given up before the step size has followed
                                * it down to dtmin; the tolerance retry of adaptive_tols treats it like any other failure) */
};

typedef struct kin_network kin_network; /* opaque */

/* ---- A1: CRN topology ------------------------------------------------------------- */
/* Replaces the four ragged vectors of RxData the solve reads (id_reacs, stoic_reacs,
 * id_prods, stoic_prods; src/exploration/network.jl:193-203) in flat CSR-like form:
 * reaction r consumes reac_idx[reac_ptr[r]..reac_ptr[r+1]) with stoichiometries reac_sto,
 * and likewise for products. ptr arrays are always 0-based offsets; idx uses index_base.
 * Rate law and ODEs are those of make_rs (src/solving/solve_utils.jl:318-334) with
 * combinatoric_ratelaws=false. */
int kin_network_create(int64_t n_species, int64_t n_reactions,
                       const int64_t* reac_ptr, const int64_t* reac_idx, const int64_t* reac_sto,
                       const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
                       int index_base, kin_network** out);
int kin_network_destroy(kin_network* h);
int kin_network_sizes(const kin_network* h, int64_t* n_species, int64_t* n_reactions);
/* Last error text of this handle (or of the failed create when h == NULL). */
const char* kin_last_error(const kin_network* h);

/* ---- A6: rate constants ------------------------------------------------------------ */
/* DummyKineticCalculator / any external calculator: hand over k[R] directly
 * (calculator.jl:127-152 produces such a vector). */
int kin_set_rates(kin_network* h, const double* k);
int kin_get_rates(kin_network* h, double* k_out);
/* PrecalculatedArrheniusCalculator parameters (calculator.jl:164-198). k_max = NaN means
 * `k_max = nothing`; t_mult = tconvert(t_unit, "s") (calculator.jl:196, utils.jl:21-30). */
int kin_set_arrhenius(kin_network* h, const double* Ea, const double* A, double k_max, double t_mult);
/* k = calculator(; T) (calculator.jl:223-232) evaluated on the device; becomes the
 * handle's current rate vector; k_out (host, R doubles) may be NULL. */
int kin_rates_at(kin_network* h, double T, double* k_out);
/* Network-free form of the same functor, used by the host for get_max_rates /
 * apply_low_k_cutoff! (solve_utils.jl:19-54, 213-245) before a handle exists. */
int kin_arrhenius_eval(const double* Ea, const double* A, int64_t n, double k_max, double t_mult,
                       double T, double* k_out);
/* A8: calculate_discrete_rates (solve_utils.jl:91-109): table[s][r] = calculator(T[s])[r],
 * S x R doubles, row-major; generated on the device. out_table (host) may be NULL, in which
 * case the table only stays resident on the device for kin_solve. */
int kin_rate_table(kin_network* h, const double* T, int64_t n_stops, double* out_table);

/* ---- A2: mass-action right-hand side ------------------------------------------------ */
/* du = f(u; k) for the current rates: the generated f! of ODEProblem (methods.jl:157). */
int kin_rhs(kin_network* h, const double* u, double* du);
/* B states at once, host buffers in the reference's layout u[b][N], du[b][N] (sol.u is a
 * Vector of Vectors); k is per state k[b][R] or NULL (current rates for every state). */
int kin_rhs_batched(kin_network* h, int64_t B, const double* u, const double* k, double* du);
/* The same sweep on device-resident buffers (no PCIe), same state-major layouts u[b][N],
 * k[b][R] (or NULL), du[b][N]. `stream` is a hipStream_t (NULL = the handle's stream); the
 * call only enqueues (no allocation, no synchronisation: graph-capturable). The handle's stream is a non-blocking
 * stream of its own: work the caller has queued on OTHER streams (fills, arithmetic on these buffers) is not ordered
 * before the sweep - pass the stream that work runs on, or synchronise first. The same holds for every *_dev entry point. */
int kin_rhs_batched_dev(kin_network* h, int64_t B, const double* d_u, const double* d_k, double* d_du, void* stream);

/* ---- A2 in the library's own data layout ("library order"): the bandwidth path of the batched sweep --------------
 * The reference evaluates its RHS one state at a time inside the integrator; the batched sweep (ensembles, flux
 * analysis over sol.u) has no layout to inherit, so the library defines one in which ONE pass over a state's rate
 * constants needs random access to on-chip memory only (kinetica_jl_amd/csrc/tiled.hpp):
 *   species   [ hubs | window 0 | window 1 | ... ]   (the caller's order when the state fits on-chip memory, N <= 10 000)
 *   reactions every reaction next to its exact reverse, grouped by window; within a window the pairs first (two slots per
 *             record, 0.0 in the reverse slot of a reaction without one), then - when the network has lost a noticeable
 *             part of its reverses, as after the low-k cutoff (solve_utils.jl:213-245) - the reactions without a reverse
 *             at one slot each: a k row has k_len <= 2 x records doubles, reaction r at slot_of_reaction[r].
 * KIN_ERR_UNSUPPORTED (from every call of this block) when the network has no such layout: more than two product
 * molecules in a reaction, or rarely referenced species that do not fall apart into window-sized groups. */
/* k_len; species_of_lib[N] (library position -> species id, + index_base); slot_of_reaction[R] (+ index_base);
 * species_identity = 1 when the library species order is the caller's; info[6] = hubs, windows, records, on-chip
 * entries, split-accumulator entries, workgroup size. Any pointer may be NULL. */
int kin_lib_layout(kin_network* h, int index_base, int64_t* k_len, int64_t* species_of_lib, int64_t* slot_of_reaction,
                   int32_t* species_identity, int64_t* info);
/* The same layout computed on the host alone (no device, no handle): what the library-order tables contain, for tools
 * and for tests that replay the sweep's arithmetic on the CPU. `hubs` = 0 lets the library choose. info[10] = hubs,
 * windows, records, on-chip entries, split-accumulator entries, workgroup size, first window entry, iteration rows,
 * k_len, 1 if some window ends in one-slot records;
 * rec[records] = packed 64-bit records (four 14-bit on-chip labels with fixed roles + 3 flag bits at bit 56),
 * rowtab[2 x rows], seg_q[windows + 1], win_off / win_cnt[windows], copy_src[copies], seg_k[2 x windows] = slot of the
 * window's first record and the number of its records that have two slots (record i of the window: slot + 2 i below
 * that number n2, slot + n2 + i from it on). Call once with NULL arrays for the sizes. */
int kin_lib_layout_host(int64_t n_species, int64_t n_reactions, const int64_t* reac_ptr, const int64_t* reac_idx,
                        const int64_t* reac_sto, const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
                        int index_base, int hubs, int64_t* info, int64_t* species_of_lib, int64_t* slot_of_reaction,
                        uint64_t* rec, int32_t* rowtab, int32_t* seg_q, int32_t* win_off, int32_t* win_cnt, int32_t* copy_src,
                        int32_t* seg_k);
/* Symbolic analysis of the Newton-matrix factorisation (I - c J; the reference's solver does this inside KLU,
 * docs/src/getting-started.md:69) WITHOUT a device: sizes only. Arguments <= 0 take the library's defaults.
 * info[0..11] = sparse pivots, dense block dimension, elimination rounds, nnz(U), nnz(L11^-1), nnz(U11^-1), nnz(L21 L11^-1),
 * nnz(U11^-1 U12), doubles per factorisation, symbolic products of the fused solve, gather-plan entries, wavefront tasks. */
int kin_lu_analyze_host(int64_t n_species, int64_t n_reactions, const int64_t* reac_ptr, const int64_t* reac_idx,
                        const int64_t* reac_sto, const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
                        int index_base, int hub_degree, int max_rounds, int max_tail_degree, int max_degree, int min_round,
                        int64_t* info);
/* Layout conversions on device buffers (each a coalesced write with a gather on the source side; enqueue only):
 * states u[b][N] caller order <-> library order, rate constants k[b][R] -> k_lib[b][k_len]. */
int kin_states_to_lib_dev(kin_network* h, int64_t B, const double* d_in, double* d_out, void* stream);
int kin_states_from_lib_dev(kin_network* h, int64_t B, const double* d_in, double* d_out, void* stream);
int kin_rates_to_lib_dev(kin_network* h, int64_t B, const double* d_k, double* d_k_lib, void* stream);
/* calculate_discrete_rates (solve_utils.jl:91-109) written directly in library order: d_out[s][k_len] (device). */
int kin_rate_table_lib_dev(kin_network* h, const double* T, int64_t n_stops, double* d_out);
/* The sweep: du_lib[b] = f(u_lib[b]; k) for b < B, device buffers in library order. Exactly one of d_k_lib
 * (k_lib[b][k_len], 16-byte aligned) and d_T (B temperatures) is non-NULL; with d_T the rate constants are formed
 * inside the sweep from the Arrhenius parameters - no k stream at all (SURVEY 8(d) M1'; the reference's continuous
 * path inlines k(T(t)) into every species ODE the same way, methods.jl:389-419 with calculator.jl:223-226).
 * HBM traffic per state: k_lib row (or nothing) + u row in, du row out. Enqueue only. */
int kin_rhs_tiled_dev(kin_network* h, int64_t B, const double* d_u_lib, const double* d_k_lib, const double* d_T,
                      double* d_du_lib, void* stream);
/* The temperature form on states in the CALLER's species order: u[b][N], T[b], du[b][N] (device). When the library
 * species order differs (species_identity == 0) the call converts the layout on the way in and out (16 N bytes per
 * state each way, workspace grown on demand). */
int kin_rhs_batched_T_dev(kin_network* h, int64_t B, const double* d_u, const double* d_T, double* d_du, void* stream);
/* The drop-in batched sweep for a caller whose RATE CONSTANTS come from this library (kin_rate_table_lib_dev, or converted once
 * by kin_rates_to_lib_dev): states u[b][N] and du[b][N] in the CALLER's species order - the reference's own sol.u layout -,
 * rate constants k_lib[b][k_len] in the slot order kin_lib_layout reports (16-byte aligned). The bandwidth-grade path at ANY
 * size and after the low-k cutoff (apply_low_k_cutoff!, solve_utils.jl:213-245, leaves reactions without their reverse):
 * hubs and windows in LDS, the k row streamed once; when the library species order differs from the caller's (N > ~10 000)
 * the states are permuted through LDS on the way in and out, one coalesced pass each (HBM traffic per state: 8 k_len + 16 N for
 * the sweep + 32 N for the two permutations). kin_rhs_batched_dev (rate constants in the caller's REACTION order) stays
 * correct at every size but cannot stream k once when the state does not fit LDS. Enqueue only. */
int kin_rhs_batched_klib_dev(kin_network* h, int64_t B, const double* d_u, const double* d_k_lib, double* d_du, void* stream);

/* ---- A3: analytic sparse Jacobian ---------------------------------------------------- */
/* Replaces ODEProblem(...; jac=true, sparse=true) (methods.jl:157-158): pattern (CSR,
 * sorted columns, diagonal always present) and values for the current rates. The reference
 * stores SparseMatrixCSC; CSR of J is CSC of transpose(J), the shim picks what it needs. */
int kin_jac_nnz(kin_network* h, int64_t* nnz);
int kin_jac_pattern(kin_network* h, int64_t* rowptr, int64_t* colidx, int index_base);
int kin_jac_values(kin_network* h, const double* u, double* vals);

/* ---- A12: ODESimulationParams (src/solving/params.jl:3-27, defaults :55-75) ---------- */
typedef struct kin_params {
  double tspan0, tspan1;     /* tspan */
  double abstol;             /* 1e-10 */
  double reltol;             /* 1e-8  */
  int32_t adaptive_tols;     /* true  */
  int32_t update_tols;       /* false */
  int32_t solve_chunks;      /* true (1): chunkwise in local time, the integrator re-initialised at every chunk start as the
                              * reference does (methods.jl:819); 0: complete timespan; 2 (EXTENSION): chunkwise with difference
                              * history, order and step size carried across chunk starts whose rate constants did not change
                              * (a StaticODESolve's chunk boundaries are not events); rate updates still re-initialise */
  int32_t ban_negatives;     /* false: isoutofdomain = any(u < 0) (methods.jl:169-171) */
  double solve_chunkstep;    /* 1e-3  */
  int64_t maxiters;          /* 100000 */
  double save_interval;      /* < 0 means `nothing` */
  double dtmin;              /* minimum step size handed to the integrator; <= 0 selects the reference's own choice:
                              * eps(solve_chunkstep) for chunkwise solves (methods.jl:232, 770), eps(tspan[end]) for
                              * complete-timespan solves (methods.jl:164, 694). A step that merely STARTS below dtmin is
                              * raised to it (CVodeSetMinStep semantics); a step pushed below it by the corrector or
                              * the error test ends the attempt with KIN_RETCODE_DTLESSTHANMIN, which the retry loop
                              * (adaptive_solve!, solve_utils.jl:376-424) answers with tighter tolerances. */
  /* `solver`, `jac`, `sparse`, `progress`, `u0`, `low_k_*`, `allow_short_u0` are consumed
   * by the host layer (the integrator is always the library's BDF with the analytic
   * sparse Jacobian). */
} kin_params;

typedef struct kin_stats {
  int64_t n_steps, n_rejected, n_rhs, n_jac, n_factor, n_linsolve, n_newton_fail;
  /* n_rhs / n_linsolve: corrector iterations the device executed (launches enqueued ahead of a decision that turn into
   * no-ops are not counted), plus the right-hand sides of restarts */
  int64_t n_chunks, n_restarts, n_retries;
  double final_abstol, final_reltol; /* what update_tols writes back (solve_utils.jl:397-401) */
  double wall_seconds;
  int64_t lu_dense_dim, lu_sparse_rows, lu_rounds, lu_nnz;
  int64_t n_lu_reused;   /* step attempts that started on a cached factorisation (the solver's LU cache) */
  int64_t lu_slots;      /* size of that cache */
  int64_t n_bad_pivot;   /* factorisations dropped because a pivot vanished (answered by a fresh Jacobian and half the step) */
  int64_t n_lu_dropped;  /* cached factorisations dropped at a restart because the Jacobian's diagonal had drifted */
} kin_stats;

/* ---- A4/A5/A9/A10: the solve --------------------------------------------------------- */
/* Integrates du/dt = f(u; k) over params->tspan from u0 with the library's variable-order
 * BDF (orders 1-5, modified Newton, analytic sparse Jacobian, on-device sparse LU), standing
 * in for init/solve!/reinit! of the user-supplied stiff solver (methods.jl:174, 241, 260,
 * 779, 819; solve_utils.jl:389) including:
 *   - chunkwise local-time solving and output stitching (methods.jl:185-303, 717-865),
 *   - complete-timespan solving (methods.jl:132-183, 655-714) when solve_chunks == 0,
 *   - discrete rate-constant updates at `tstops` (solve_utils.jl:435-509): k is held
 *     piecewise constant and switched at every tstop; rates come from k_table[s][R] (host,
 *     any calculator) or, when k_table == NULL, from the Arrhenius parameters at T_stops[s];
 *     n_stops == 0 means static rates (StaticODESolve),
 *   - the tolerance-tightening retry loop adaptive_solve! (solve_utils.jl:376-424).
 * Results stay in the handle; fetch them with kin_solution_size / kin_solution_copy.
 * Returns KIN_ERR_SOLVE_FAILED (with *retcode set) when adaptive_solve! would throw. */
int kin_solve(kin_network* h, const kin_params* params, const double* u0,
              const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops,
              int64_t* n_saved, int32_t* retcode, kin_stats* stats);
/* An ENSEMBLE of K independent trajectories of one network in ONE call (the reference leaves ensembles to the user:
 * docs/src/tutorials/ode-solution.md:190 solves member after member through solve_network, exploration/methods.jl:221).
 * Every member is integrated by the same algorithm as kin_solve (csrc/resident_core.hpp is the controller of all paths):
 *   - a network that fits one compute unit's LDS (up to ~1 000 species): ONE launch, one workgroup owns one member from u0 to
 *     the end of the span; a member's result is bit-identical to a K = 1 call with its inputs;
 *   - larger networks, up to 12 members (KIN_ENSEMBLE_THREADS): K kin_solve calls on K host threads, each on a solve-only copy
 *     of the handle (kept with the handle for later calls); bit-identical to kin_solve on the member's inputs;
 *   - larger networks, more members: the members advance in lockstep rounds, every launch of a round carries all members
 *     that need that kind of work (csrc/ensemble.cpp); a member's result equals its solo kin_solve within the step-sequence
 *     tolerance.
 *   u0[K][N]; rate constants per member k[K][R], or temperatures T[K] (Arrhenius parameters of the handle), or neither
 *   (the handle's current rates for all); discrete rate updates (tstops / T_stops / k_table as in kin_solve) are shared
 *   by all members and exclude k / T. `params` needs a save grid (solve_chunks or save_interval).
 * Outputs (any may be NULL): *n_rows = rows of the save grid; out_t[n_rows]; out_u[K][n_rows][N]; n_saved[K] rows a member
 * actually wrote (the rows of out_u beyond them - a member that failed early - are zero); retcodes[K] (KIN_RETCODE_*); stats[K].
 * A call with out_u == NULL and n_saved == NULL only reports *n_rows.
 * Returns KIN_OK when the call ran, whatever the members' retcodes; KIN_ERR_UNSUPPORTED for a network that fits neither
 * path (too large for the resident kernel and without a dense Schur block): solve those member by member with kin_solve. */
int kin_solve_ensemble(kin_network* h, const kin_params* params, int64_t K, const double* u0, const double* k, const double* T,
                       const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* n_rows,
                       double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats);
/* Same call with an explicit integrator: `pars.solver` may be any SciML algorithm (params.jl:9); BASELINE config 2
 * exercises the RHS kernel with an explicit one. The pair is Dormand-Prince 5(4) with FSAL and 4th-order dense
 * output, step-size control as in SciPy's RK45 (which is the oracle, step for step); no Jacobian, no linear solve.
 * Orchestration (chunks, save grid, discrete rate updates, tolerance retries) is kin_solve's. */
int kin_solve_explicit(kin_network* h, const kin_params* params, const double* u0, const double* tstops, const double* T_stops,
                       const double* k_table, int64_t n_stops, int64_t* n_saved, int32_t* retcode, kin_stats* stats);

/* N3: continuous rate updates (reference: methods.jl:363-653, where k(t) = calculator(T(t)) is inlined
 * symbolically into every species ODE). Here the integrator is simply non-autonomous: the Arrhenius
 * rates are re-evaluated on the device at T(t_new) for every step attempt; T(t) is the linear
 * interpolation of the profile solution (t_nodes, T_nodes), as the reference's DiffEqArray functor does
 * (src/utils.jl:135-139). Chunking / save grid / retry semantics as kin_solve; no restarts at all. */
int kin_solve_continuous(kin_network* h, const kin_params* params, const double* u0, const double* t_nodes,
                         const double* T_nodes, int64_t n_nodes, int64_t* n_saved, int32_t* retcode, kin_stats* stats);
/* N1: return_integrator=true (methods.jl:105-106, 175-178, 242-246, 706-709): `init(oprob, solver; kwargs...)`
 * without solve!. The integrator spans the whole tspan (solve_chunks = 0) or the first chunk
 * [0, solve_chunkstep] (solve_chunks = 1, what the reference hands back); tstops / T_stops / k_table as
 * in kin_solve (n_stops = 0: the rates set on the handle). Save grid and the tolerance retry loop are
 * not part of an integrator (they belong to adaptive_solve!, solve_utils.jl:376-424). */
int kin_integrator_init(kin_network* h, const kin_params* params, const double* u0, const double* tstops,
                        const double* T_stops, const double* k_table, int64_t n_stops);
/* The same for continuous rate updates (methods.jl:363-458 with :445-449, and :461-653): the integrator re-evaluates
 * the Arrhenius rates at T(t) of every step attempt, T(t) as in kin_solve_continuous. */
int kin_integrator_init_continuous(kin_network* h, const kin_params* params, const double* u0, const double* t_nodes,
                                   const double* T_nodes, int64_t n_nodes);
/* step!(integ) x max_steps accepted steps (max_steps <= 0: solve!(integ), run to the end of the span);
 * rate updates fire when the time reaches a tstop (solve_utils.jl:435-509). steps_taken < max_steps
 * means the end of the span was reached or the integrator failed (see kin_integrator_state). */
int kin_integrator_step(kin_network* h, int64_t max_steps, int64_t* steps_taken);
/* integ.t, integ.u[N], the integrator's KIN_RETCODE_* and counters; any pointer may be NULL. */
int kin_integrator_state(kin_network* h, double* t, double* u, int32_t* retcode, kin_stats* stats);
int kin_solution_size(const kin_network* h, int64_t* n_saved, int64_t* n_species);
/* out_t[n_saved], out_u[n_saved][N] (sol.t / sol.u of ODESolveOutput, analysis/io.jl:3-11). */
int kin_solution_copy(const kin_network* h, double* out_t, double* out_u);
/* N2: max over saved times of each species (what identify_next_seeds reads,
 * src/exploration/explore_utils.jl:344-351), reduced on the device. */
int kin_solution_max(const kin_network* h, double* out_umax);

/* ---- multi-GPU building blocks (one process per GPU; the collectives themselves are RCCL calls of the host layer) -- */
/* kin_solution_max into a caller-owned DEVICE buffer of N doubles: what an ensemble of replicas all-gathers. */
int kin_solution_max_dev(const kin_network* h, double* d_out);
/* kin_rate_table for a slice of time stops straight into a caller-owned DEVICE buffer [n_stops][R] (T on the host):
 * ranks generate disjoint slices of the table; an all-gather follows only if sol_k is wanted. */
int kin_rate_table_dev(kin_network* h, const double* T, int64_t n_stops, double* d_out);
/* Partial right-hand side of reactions [r_lo, r_hi) only: du_partial = sum over the block of nu * rate (device buffers,
 * enqueue only). Summed over a partition of the reactions (an all-reduce of N doubles) it is kin_rhs: the reaction-block
 * decomposition of ONE trajectory. */
int kin_rhs_block_dev(kin_network* h, int64_t r_lo, int64_t r_hi, const double* d_u, double* d_du, void* stream);

/* out[t] = sum_i w[i] * u_i(t) for every saved time, reduced on the device: conserved quantities of the network
 * (element or mass balances - what a caller checks before trusting a long run) without copying the trajectory. */
int kin_solution_dot(const kin_network* h, const double* w, double* out);
/* Selected rows of the device-resident rate table (k_precalc[s] of calculate_discrete_rates, solve_utils.jl:91-109):
 * out[i][R] = table[rows[i]][:]. The full table (5.6 GB at 14 001 stops x 50 000 reactions) never has to cross PCIe
 * for a caller that wants sol_k at a few stops. */
int kin_rate_table_rows(kin_network* h, const int64_t* rows, int64_t n_rows, double* out);

/* Diagnostic: one Newton-matrix solve on the device, (I - c*J(u)) x = b with the current rates,
 * through exactly the factorisation / substitution kernels kin_solve uses (what KLU does for
 * CVODE in the reference's documented setup, docs/src/getting-started.md:69). */
int kin_newton_solve(kin_network* h, double c, const double* u, const double* b, double* x);

/* ---- device / build information ------------------------------------------------------- */
int kin_device_count(int* n);
/* Selects the device for handles created afterwards by this thread; a handle remembers the device it was created on
 * and every later call on it runs there, whatever the calling thread's current device is. Handles are independent:
 * K handles driven by K host threads may share one GPU (concurrent replicas of an ensemble) or sit on different ones. */
int kin_set_device(int device);
const char* kin_version(void);
/* Layout version of this header's structs and argument lists. A binding compares it (and, if it wants certainty, the
 * struct sizes) with the values it was written against before the first call: kin_params / kin_stats have grown between
 * versions (1: round 1; 2: + dtmin and the LU-cache counters; 3: + the library-order sweep entry points; 4: + kin_solve_ensemble,
 * kin_lu_analyze_host - structs unchanged; 5: + kin_rhs_batched_klib_dev - structs unchanged). */
#define KIN_ABI_VERSION 5
int kin_abi_version(void);
int64_t kin_struct_size(int which); /* 0: sizeof(kin_params), 1: sizeof(kin_stats), else -1 */

#ifdef __cplusplus
}
#endif
#endif /* KINETICA_HIP_H */
