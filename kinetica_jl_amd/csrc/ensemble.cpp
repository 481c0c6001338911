// Batched ensemble of large networks: see ensemble.hpp. One host thread per member runs the controller of
// resident_core.hpp; its backend (MemberBackend) hands every operation to the ROUND: when all members still running have
// handed one in, the last arriver launches the round - one batched launch sequence per kind of operation present, the
// members' factorisations spread over a few streams next to it - waits for the device once, and releases the others.
#include "ensemble.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <deque>
#include <thread>

#include "handle.hpp"
#include "lu.hpp"
#include "resident_setup.hpp"
#include "solver.hpp"

namespace kin {

namespace {

enum OpKind : int { K_NONE = 0, K_VEC, K_APPLY_RATES, K_RHS, K_JAC, K_NORMS, K_INIT_D, K_DRIFT, K_FACTOR, K_CORRECTOR, K_CORRECTOR_CONT, K_KINDS };
constexpr int BLIND_ITERS = 2;   // corrector iterations enqueued ahead of the decision (most attempts are decided by then; the rest continue next round)

struct Pending {
  int kind = K_NONE;
  EnsOp op{};
  bool pre_accept = false, pre_change = false;
  int accept_order = 0;
  double pre_ru[36];
  // factor
  // drift
  double max_drift = 0.0;
  // results
  bool bad = false;
  int dropped = 0;
};

}  // namespace

struct EnsembleSolver {
  kin_network* h;
  SparseLU lu;
  SegPlanDev resid_plan;
  DevBuf<int32_t> d_jdiag;
  EnsSolveTables T{};
  bool ok = false;
  std::string why;
  int N = 0, R = 0;
  int64_t nnz = 0;
  // per solve
  int K = 0, n_slots = 0;
  int64_t cap = 0;
  DevBuf<double> state, sol, d_drift;
  DevBuf<BdfCtrl> d_ctrl;
  DevBuf<int> d_bad;
  DevBuf<EnsRep> d_reps;
  DevBuf<EnsOp> d_ops;
  std::vector<EnsRep> reps;
  BdfCtrl* h_ctrl = nullptr;      // pinned (coherent, mapped), K
  BdfCtrl* h_ctrl_dev = nullptr;  // ... its device address
  unsigned long long* h_seq = nullptr; unsigned long long* h_seq_dev = nullptr;   // sequence number of the last published round
  unsigned long long seq_no = 0;
  bool fast_sync = true;
  int64_t n_fast = 0, n_slow = 0;
  int* h_bad = nullptr;           // pinned, K
  EnsOp* h_ops = nullptr;         // pinned, 3 K (pre-accepts | pre-changes | the round's operations by kind)
  double* h_drift = nullptr;      // pinned, K x LU_MAX_SLOTS
  std::vector<std::vector<SparseLU::Slot>> slots;   // [member][slot]
  std::vector<std::vector<double>> sol_t;           // [member] save times
  // the round
  std::mutex mu;
  std::condition_variable cv;
  int active = 0, arrived = 0;
  uint64_t epoch = 0;
  std::vector<Pending> pend, results;
  std::string round_error;
  int64_t n_rounds = 0;
  // A factorisation is a chain of ~40 chip-wide launches (0.5 ms): it runs on a stream of its own and its member is PARKED -
  // the rounds go on without it and pick it up again when its event has completed
  std::vector<char> done;
  // a small pool of streams for the members' factorisations (member t uses stream t mod pool; a chain holds the stream's lock
  // while it is being enqueued: with a stream per member the chip's few hardware queues multiplex dozens of streams and
  // the rounds' own launches wait behind them - measured 16.6 solves/s at K = 16 against 18-20 with a pool)
  std::vector<hipStream_t> ms;
  std::vector<DevBuf<double>> mpinv;    // ... each with its scratch for the dense inverse's pivot blocks
  std::vector<std::unique_ptr<std::mutex>> ms_lock;
  std::vector<hipEvent_t> evs;          // per member: end of its factorisation chain
  int pool = 8;
  int n_parked = 0;
  // The dense inverses of members that factorise at the same time run as ONE chain of launches (launch_gauss_jordan_batched:
  // 157 us per matrix at 8 matrices against 460 us alone, tools/gj_probe.hip): a member's thread enqueues the sparse part of its
  // factorisation on its pool stream and hands the dense block to this server; the server takes what has queued up while its
  // previous batch was running (that wait IS the batching window; plus a short hold-off), at most GJ_BMAX. Measured at C3 size
  // (tools/ens_gj_matrix.sh, best of 4): K = 16: 18.1 solves/s against 16.6-17.6 with every member's own chain, K = 32: 20.4-20.7
  // against 18.4-18.9; 2.6-3.1 matrices per chain. The device is throughput-bound on the members' summed kernel time by then.
  struct GjReq { int t; SparseLU::Slot* q; hipEvent_t ready; };
  std::mutex gmu;
  std::condition_variable gcv, gcv_members;
  std::deque<GjReq> gq;
  std::vector<char> g_enqueued;         // per member: its batch is in the stream and evs[t] recorded behind it
  std::string g_error;
  bool g_stop = false, g_batched = true;
  std::thread g_thread;
  hipStream_t gs = nullptr;
  DevBuf<double> gpinv;
  std::vector<hipEvent_t> pre_evs;      // per member: end of the sparse part
  int64_t g_batches = 0, g_matrices = 0;
  // (factorisations kept INSIDE the rounds - every launch of a round then carries all members - were built in round 4 and measured
  // slower, profiles/r04_ensemble_factor_sync_ab.txt; that path is gone)
  int g_min = 4, g_wait_us = 200;       // hold-off: a batch starts with 4 requests or 200 us after its first
  double t_enqueue = 0.0, t_sync = 0.0, t_round = 0.0;   // host seconds inside the rounds (KIN_TIMING=1)
  int64_t n_ops[16] = {};

  explicit EnsembleSolver(kin_network* hh) : h(hh) {
    const NetworkHost& H = h->host;
    hipStream_t s = h->stream;
    N = (int)H.N; R = (int)H.R; nnz = H.nnz();
    LUOptions opt;
    if (N >= 4000) { opt.max_tail_degree = 32; opt.max_rounds = 16; opt.max_degree = 400; }   // as Solver (solver.cpp)
    lu.analyze(N, H.j_ptr, H.j_col, opt, s);
    lu.slots.clear();
    if (!(lu.fused_tri && lu.m > 0)) { why = "the batched ensemble needs the fused solve form (a network with a dense Schur block)"; return; }
    std::vector<int32_t> yl(N), ident(N);
    lu.yloc.download(yl.data(), N, s);
    KIN_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) ident[i] = i;
    resid_plan.upload(build_seg_plan(N, H.sp_ptr.data(), yl.data(), H.sp_rxn.data(), nullptr, H.sp_coef.data(), false, ident.data()), s);
    d_jdiag.upload(H.j_diag, s);
    T.N = N; T.R = R; T.m = lu.m; T.mpad = lu.mpad; T.ns = lu.ns; T.off_y = lu.off_y; T.off_x = lu.off_x;
    T.x0 = h->x0.p; T.x1 = h->x1.p; T.xloc = lu.xloc.p; T.x2_species = lu.x2_species.p;
    T.resid = resid_plan.view(); T.stageA = lu.stageA.view(); T.stageC = lu.stageC.view();
    const double KAPPA[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    T.cf.gamma[0] = 0.0;
    for (int j = 1; j <= BDF_MAX_ORDER; j++) T.cf.gamma[j] = T.cf.gamma[j - 1] + 1.0 / j;
    for (int j = 0; j <= BDF_MAX_ORDER; j++) T.cf.alpha[j] = (1.0 - KAPPA[j]) * T.cf.gamma[j];
    for (int j = 0; j <= BDF_MAX_ORDER; j++) T.cf.error_const[j] = KAPPA[j] * T.cf.gamma[j] + 1.0 / (j + 1);
    T.cf.error_const[BDF_MAX_ORDER + 1] = 0.0;
    ok = true;
  }
  ~EnsembleSolver() {
    stop_gj_server();
    if (gs) (void)hipStreamDestroy(gs);
    for (auto& e : pre_evs) if (e) (void)hipEventDestroy(e);
    for (auto& m_ : ms) if (m_) (void)hipStreamDestroy(m_);
    for (auto& e : evs) if (e) (void)hipEventDestroy(e);
    if (h_ctrl) (void)hipHostFree(h_ctrl);
    if (h_seq) (void)hipHostFree(h_seq);
    if (h_bad) (void)hipHostFree(h_bad);
    if (h_ops) (void)hipHostFree(h_ops);
    if (h_drift) (void)hipHostFree(h_drift);
  }

  void prepare(int K_, int slots_, int64_t cap_) {
    hipStream_t s = h->stream;
    auto al = [](size_t x) { return (x + 7) / 8 * 8; };
    const size_t n = (size_t)N, r = (size_t)R;
    const size_t part = (size_t)ens_reduce_doubles(N);
    const size_t per = al((size_t)BDF_D_ROWS * n) + 8 * al(n) + al((size_t)nnz) + al(r) + al(2 * r + 2) + al(r) + al(part);
    if (K_ != K) {
      // (freed pointers are nulled and K forgotten first: an allocation below may throw, and neither the destructor nor the
      // next call may find a dangling pointer or a K that matches buffers which are gone)
      if (h_ctrl) (void)hipHostFree(h_ctrl);
      if (h_bad) (void)hipHostFree(h_bad);
      if (h_ops) (void)hipHostFree(h_ops);
      if (h_drift) (void)hipHostFree(h_drift);
      h_ctrl = nullptr; h_bad = nullptr; h_ops = nullptr; h_drift = nullptr; h_ctrl_dev = nullptr;
      K = 0;
      KIN_HIP(hipHostMalloc((void**)&h_ctrl, (size_t)K_ * sizeof(BdfCtrl), hipHostMallocCoherent | hipHostMallocMapped));
      if (!h_seq) { KIN_HIP(hipHostMalloc((void**)&h_seq, sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped)); *h_seq = 0; }
      fast_sync = !getenv("KIN_NO_FAST_SYNC");
      if (hipHostGetDevicePointer((void**)&h_ctrl_dev, h_ctrl, 0) != hipSuccess || hipHostGetDevicePointer((void**)&h_seq_dev, h_seq, 0) != hipSuccess) {
        (void)hipGetLastError();
        fast_sync = false;
      }
      KIN_HIP(hipHostMalloc((void**)&h_bad, (size_t)K_ * sizeof(int), hipHostMallocDefault));
      KIN_HIP(hipHostMalloc((void**)&h_ops, (size_t)3 * K_ * sizeof(EnsOp), hipHostMallocDefault));
      KIN_HIP(hipHostMalloc((void**)&h_drift, (size_t)K_ * LU_MAX_SLOTS * sizeof(double), hipHostMallocDefault));
    }
    K = K_; n_slots = slots_; cap = cap_;
    state.alloc((size_t)K * per);
    sol.alloc((size_t)K * (size_t)cap * n);
    // rows beyond a member's n_saved (a member that failed early) read as zeros in out_u, not as whatever the buffer held
    KIN_HIP(hipMemsetAsync(sol.p, 0, (size_t)K * (size_t)cap * n * sizeof(double), s));
    d_ctrl.alloc(K); d_bad.alloc(K); d_reps.alloc(K); d_ops.alloc((size_t)3 * K); d_drift.alloc((size_t)K * LU_MAX_SLOTS);
    KIN_HIP(hipMemsetAsync(d_ctrl.p, 0, (size_t)K * sizeof(BdfCtrl), s));
    KIN_HIP(hipMemsetAsync(d_bad.p, 0, (size_t)K * sizeof(int), s));
    reps.assign(K, EnsRep{});
    for (int t = 0; t < K; t++) {
      EnsRep& q = reps[t];
      double* w = state.p + (size_t)t * per;
      q.D = w; w += al((size_t)BDF_D_ROWS * n);
      q.y = w; w += al(n); q.psi = w; w += al(n); q.d = w; w += al(n); q.scale = w; w += al(n);
      q.f0 = w; w += al(n); q.f1 = w; w += al(n); q.ytmp = w; w += al(n); q.cs = w; w += al(n);
      q.jv = w; w += al((size_t)nnz);
      q.rate = w; w += al(r);
      q.dr = w; w += al(2 * r + 2);
      q.k = w; w += al(r);
      q.part = w;
      q.ctrl = d_ctrl.p + t;
    }
    d_reps.upload(reps.data(), (size_t)K, s);
    slots.resize(K);
    for (auto& v : slots) { if ((int)v.size() < n_slots) v.resize(n_slots); for (auto& q : v) q.valid = false; }
    sol_t.assign(K, std::vector<double>((size_t)cap, 0.0));
    pend.assign(K, Pending{});
    results.assign(K, Pending{});
    done.assign(K, 0); n_parked = 0;
    ctrl_of.assign(K, BdfCtrl{});
    if (const char* e = getenv("KIN_ENSEMBLE_STREAMS")) pool = std::max(1, atoi(e));
    while ((int)evs.size() < K) { hipEvent_t e; KIN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); evs.push_back(e); }
    while ((int)pre_evs.size() < K) { hipEvent_t e; KIN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); pre_evs.push_back(e); }
    if (!gs) KIN_HIP(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
    gpinv.alloc((size_t)GJ_BMAX * 2 * 32 * 32);
    g_enqueued.assign(K, 0);
    g_batched = !(getenv("KIN_ENSEMBLE_GJ_BATCHED") && atoi(getenv("KIN_ENSEMBLE_GJ_BATCHED")) == 0);
    g_batches = g_matrices = 0;
    g_min = 4; g_wait_us = 200;
    g_min = std::min(g_min, std::max(1, K / 4));       // a small ensemble does not hold a lone request back for partners that rarely come
    while ((int)ms.size() < pool) {
      hipStream_t m_; KIN_HIP(hipStreamCreateWithFlags(&m_, hipStreamNonBlocking));
      ms.push_back(m_); mpinv.emplace_back(); mpinv.back().alloc(2 * 32 * 32); ms_lock.emplace_back(new std::mutex());
    }
    active = K; arrived = 0; epoch = 0; round_error.clear(); n_rounds = 0;
    t_enqueue = t_sync = t_round = 0.0; n_fast = n_slow = 0;
    for (auto& x : n_ops) x = 0;
    KIN_HIP(hipStreamSynchronize(s));
  }

  // ---- the round (called with `mu` held by the last arriver)
  void execute_round() {
    n_rounds++;
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr1 = tr0;
    for (int t = 0; t < K; t++) n_ops[pend[t].kind & 15]++;
    try {
      hipStream_t s = h->stream;
      KIN_HIP(hipSetDevice(h->device));
      // 1. the operations every member deferred (accept of its last step, step-size changes) go first, in that order
      int na = 0, nc = 0;
      EnsOp* A = h_ops;
      EnsOp* C = h_ops + K;
      EnsOp* O = h_ops + 2 * K;
      for (int t = 0; t < K; t++) {
        Pending& p = pend[t];
        if (p.kind == K_NONE) continue;
        if (p.pre_accept) { A[na] = EnsOp{}; A[na].rep = t; A[na].i0 = p.accept_order; na++; }
        if (p.pre_change) { C[nc] = EnsOp{}; C[nc].rep = t; std::copy(p.pre_ru, p.pre_ru + 36, C[nc].ru); nc++; }
      }
      // 2. the operations themselves, grouped by kind
      int first[K_KINDS], cnt[K_KINDS];
      int no = 0;
      for (int kind = 0; kind < K_KINDS; kind++) {
        first[kind] = no; cnt[kind] = 0;
        if (kind == K_NONE || kind == K_DRIFT) continue;
        for (int t = 0; t < K; t++) if (pend[t].kind == kind) { O[no] = pend[t].op; O[no].rep = t; no++; cnt[kind]++; }
      }
      auto count = [&](int kind) { return cnt[kind]; };
      if (na + nc + no > 0) KIN_HIP(hipMemcpyAsync(d_ops.p, h_ops, (size_t)3 * K * sizeof(EnsOp), hipMemcpyHostToDevice, s));
      ens_accept(N, d_reps.p, d_ops.p, na, s);
      ens_change_D(N, d_reps.p, d_ops.p + K, nc, s);
      const EnsOp* dO = d_ops.p + 2 * K;
      ens_vec(N, d_reps.p, dO + first[K_VEC], count(K_VEC), s);
      ens_apply_rates(R, h->Ea.p, h->A.p, h->has_kmax ? 1 : 0, h->k_max, h->t_mult, d_reps.p, dO + first[K_APPLY_RATES], count(K_APPLY_RATES), s);
      ens_rhs(N, R, h->x0.p, h->x1.p, h->rhs_plan.view(), d_reps.p, dO + first[K_RHS], count(K_RHS), s);
      ens_jac(R, h->x0.p, h->x1.p, h->jac_plan.view(), d_reps.p, dO + first[K_JAC], count(K_JAC), s);
      ens_norms(N, d_reps.p, dO + first[K_NORMS], count(K_NORMS), s);
      ens_init_D(N, d_reps.p, dO + first[K_INIT_D], count(K_INIT_D), s);
      const int ncorr = count(K_CORRECTOR), ncont = count(K_CORRECTOR_CONT);
      if (ncorr > 0) {
        ens_predict(T, d_reps.p, dO + first[K_CORRECTOR], ncorr, s);
        ens_iterations(T, d_reps.p, dO + first[K_CORRECTOR], ncorr, 0, BLIND_ITERS, s);
      }
      if (ncont > 0) ens_iterations(T, d_reps.p, dO + first[K_CORRECTOR_CONT], ncont, BLIND_ITERS, BDF_NEWTON_MAXITER - BLIND_ITERS, s);
      // 3. drift checks (one small launch per member at a restart)
      bool any_drift = false;
      for (int t = 0; t < K; t++) {
        Pending& p = pend[t];
        if (p.kind == K_DRIFT) {
          SlotDriftArgs a;
          const int ns = (int)slots[t].size();
          for (int i = 0; i < ns; i++) { a.jd[i] = slots[t][i].valid ? slots[t][i].jd.p : nullptr; a.c[i] = slots[t][i].c_fact; }
          launch_slot_drift(N, ns, reps[t].jv, d_jdiag.p, a, d_drift.p + (size_t)t * LU_MAX_SLOTS, s);
          any_drift = true;
        }
      }
      if (any_drift) KIN_HIP(hipMemcpyAsync(h_drift, d_drift.p, (size_t)K * LU_MAX_SLOTS * sizeof(double), hipMemcpyDeviceToHost, s));
      // the round's hand-over: the control blocks published into pinned host memory by the round's last launch and a sequence
      // number to spin on (the host-driven integrator's scheme; ~5 us instead of the ~25 us of a copy + stream synchronisation),
      // unless other results come back by copy in this round (drift tests, factorisation flags) or the number does not arrive
      bool waited = false;
      if (fast_sync && !any_drift) {
        const unsigned long long want = ++seq_no;
        ens_publish(d_ctrl.p, h_ctrl_dev, K, h_seq_dev, want, s);
        tr1 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
          if (*(volatile unsigned long long*)h_seq >= want) { std::atomic_thread_fence(std::memory_order_acquire); waited = true; n_fast++; break; }
          __builtin_ia32_pause();
          if ((spins & 4095) == 4095 && std::chrono::duration<double>(std::chrono::steady_clock::now() - tr1).count() > 50e-3) { fast_sync = false; break; }
        }
      }
      if (!waited) {
        KIN_HIP(hipMemcpyAsync(h_ctrl, d_ctrl.p, (size_t)K * sizeof(BdfCtrl), hipMemcpyDeviceToHost, s));
        tr1 = std::chrono::steady_clock::now();
        KIN_HIP(hipStreamSynchronize(s));
        n_slow++;
      }
      t_sync += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr1).count();
      for (int t = 0; t < K; t++) {
        Pending& p = pend[t];
        if (p.kind == K_NONE) continue;
        if (p.kind == K_DRIFT) {
          p.dropped = 0;
          for (int i = 0; i < (int)slots[t].size(); i++)
            if (slots[t][i].valid && !(h_drift[(size_t)t * LU_MAX_SLOTS + i] <= p.max_drift)) { slots[t][i].valid = false; p.dropped++; }
        }
        ctrl_of[t] = h_ctrl[t];
        results[t] = p;
        done[t] = 1;
      }
    } catch (const std::exception& e) {
      round_error = e.what();
    }
    t_enqueue += std::chrono::duration<double>(tr1 - tr0).count();
    t_round += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
  }

  std::vector<BdfCtrl> ctrl_of;   // every member's control block as of its own last operation
  // the last arriver runs the round, publishes its members' results and releases them
  void finish_round() {
    execute_round();
    for (auto& p : pend) p.kind = K_NONE;
    arrived = 0;
    if (!round_error.empty()) for (int t = 0; t < K; t++) done[t] = 1;
    epoch++;
    cv.notify_all();
  }
  // hands member t's operation to the round and waits for its completion; returns the member's entry with its results
  Pending run_op(int t, const Pending& p) {
    std::unique_lock<std::mutex> lk(mu);
    pend[t] = p;
    done[t] = 0;
    arrived++;
    if (arrived >= active - n_parked) finish_round();
    cv.wait(lk, [&] { return done[t] != 0; });
    if (!round_error.empty()) throw KinError(ERR_DEVICE, "ensemble round failed: " + round_error);
    return results[t];
  }
  // A factorisation is a chain of ~45 chip-wide launches (0.5 ms): the member's OWN thread enqueues it on the member's own
  // stream and waits for it there; meanwhile the rounds go on without the member (`away`), and the launch calls of several
  // members' chains proceed in parallel instead of queueing up in the round's executor
  void away() {
    std::unique_lock<std::mutex> lk(mu);
    n_parked++;
    if (arrived > 0 && arrived >= active - n_parked) finish_round();
  }
  void back() {
    std::unique_lock<std::mutex> lk(mu);
    n_parked--;
  }
  bool factor_member(int t, int slot, double c, bool keep_diag) {
    away();
    bool bad = false;
    std::string err;
    try {
      KIN_HIP(hipSetDevice(h->device));
      const int si = t % pool;
      hipStream_t f = ms[si];
      hipEvent_t done_ev = evs[t];
      SparseLU::Slot& q = slots[t][slot];
      const bool batched = g_batched && lu.m > 0;
      {
        std::lock_guard<std::mutex> g(*ms_lock[si]);
        if (!q.W.p) lu.alloc_slot(q, f);
        if (batched) lu.factor_sparse_into(c, reps[t].jv, q, d_bad.p + t, f);
        else lu.factor_into(c, reps[t].jv, q, mpinv[si].p, d_bad.p + t, f);
        if (keep_diag) { q.jd.alloc(N); launch_jac_diag(N, reps[t].jv, d_jdiag.p, q.jd.p, f); }
        if (batched) KIN_HIP(hipEventRecord(pre_evs[t], f));
        else {
          KIN_HIP(hipMemcpyAsync(h_bad + t, d_bad.p + t, sizeof(int), hipMemcpyDeviceToHost, f));
          KIN_HIP(hipMemsetAsync(d_bad.p + t, 0, sizeof(int), f));
          KIN_HIP(hipEventRecord(done_ev, f));
        }
      }
      if (batched) {
        std::unique_lock<std::mutex> lk(gmu);
        g_enqueued[t] = 0;
        gq.push_back(GjReq{t, &q, pre_evs[t]});
        gcv.notify_one();
        gcv_members.wait(lk, [&] { return g_enqueued[t] != 0 || !g_error.empty(); });
        if (!g_error.empty()) throw KinError(ERR_DEVICE, g_error);
      }
      KIN_HIP(hipEventSynchronize(done_ev));
      bad = h_bad[t] != 0;
    } catch (const std::exception& e) { err = e.what(); }
    back();
    if (!err.empty()) throw KinError(ERR_DEVICE, "ensemble factorisation failed: " + err);
    return bad;
  }
  void gj_server() {
    (void)hipSetDevice(h->device);
    std::vector<GjReq> batch;
    for (;;) {
      batch.clear();
      {
        std::unique_lock<std::mutex> lk(gmu);
        gcv.wait(lk, [&] { return g_stop || !gq.empty(); });
        if (gq.empty()) return;          // stop requested and nothing left
        // hold-off: members of a lockstep ensemble tend to factorise within a few rounds of each other
        if ((int)gq.size() < g_min && g_wait_us > 0)
          gcv.wait_for(lk, std::chrono::microseconds(g_wait_us), [&] { return g_stop || (int)gq.size() >= g_min; });
        while (!gq.empty() && (int)batch.size() < GJ_BMAX) { batch.push_back(gq.front()); gq.pop_front(); }
      }
      std::string err;
      try {
        const int n = (int)batch.size();
        double* S[GJ_BMAX]; double* S2[GJ_BMAX]; int* bad[GJ_BMAX];
        for (int i = 0; i < n; i++) {
          KIN_HIP(hipStreamWaitEvent(gs, batch[i].ready, 0));
          S[i] = batch[i].q->W.p + lu.off_S; S2[i] = batch[i].q->S2.p; bad[i] = d_bad.p + batch[i].t;
        }
        const int where = launch_gauss_jordan_batched(n, S, S2, lu.mpad, gpinv.p, bad, gs);
        for (int i = 0; i < n; i++) {
          const int t = batch[i].t;
          batch[i].q->sinv = where ? S2[i] : S[i];
          KIN_HIP(hipMemcpyAsync(h_bad + t, d_bad.p + t, sizeof(int), hipMemcpyDeviceToHost, gs));
          KIN_HIP(hipMemsetAsync(d_bad.p + t, 0, sizeof(int), gs));
          KIN_HIP(hipEventRecord(evs[t], gs));
        }
        g_batches++; g_matrices += n;
      } catch (const std::exception& e) { err = e.what(); }
      {
        std::lock_guard<std::mutex> lk(gmu);
        if (!err.empty()) g_error = "dense-inverse server: " + err;
        for (auto& r : batch) g_enqueued[r.t] = 1;
      }
      gcv_members.notify_all();
      if (err.empty()) (void)hipEventSynchronize(evs[batch.back().t]);   // requests that arrive meanwhile form the next batch
    }
  }
  void start_gj_server() {
    stop_gj_server();
    { std::lock_guard<std::mutex> lk(gmu); g_stop = false; g_error.clear(); gq.clear(); }
    if (g_batched) g_thread = std::thread([this] { gj_server(); });
  }
  void stop_gj_server() {
    if (!g_thread.joinable()) return;
    { std::lock_guard<std::mutex> lk(gmu); g_stop = true; }
    gcv.notify_all();
    g_thread.join();
  }
  // a member that has finished no longer takes part; the round it was the last one missing from runs now
  void member_done() {
    std::unique_lock<std::mutex> lk(mu);
    active--;
    if (arrived > 0 && arrived >= active - n_parked) finish_round();
  }
};

void EnsembleDeleter::operator()(EnsembleSolver* p) const { delete p; }

namespace {

// The backend of ONE member's controller (resident_core.hpp: ResidentBdf<MemberBackend>): vector operations, right-hand
// sides, Jacobians, factorisations and corrector attempts become entries of the ensemble's rounds; the slot table of the
// member's LU cache and its deferred operations live here, on the host.
struct MemberBackend {
  EnsembleSolver& E;
  const int t;
  const ResParams& P;
  const double* u0_dev;       // this member's initial state (device)
  const double* table_dev;    // shared rate table (device) or null
  const double* T_stops;      // shared temperatures of the stops (host) or null
  // deferred operations on the difference array D: the accept of the last step, then step-size changes (composed)
  bool pend_accept = false, pend_change = false;
  int pend_accept_order = 0;
  double pend_ru[36];
  int64_t n_saved_rows = 0;

  MemberBackend(EnsembleSolver& e, int t_, const ResParams& p, const double* u0d, const double* tab, const double* Ts)
      : E(e), t(t_), P(p), u0_dev(u0d), table_dev(tab), T_stops(Ts) {}

  int n_species() const { return E.N; }
  void profile_out(int64_t*) const {}

  Pending make(int kind) {
    Pending p;
    p.kind = kind;
    p.pre_accept = pend_accept; p.accept_order = pend_accept_order;
    p.pre_change = pend_change;
    if (pend_change) std::copy(pend_ru, pend_ru + 36, p.pre_ru);
    pend_accept = false; pend_change = false;
    return p;
  }
  Pending run(const Pending& p) { return E.run_op(t, p); }

  // ---- slot table
  std::vector<SparseLU::Slot>& sl() { return E.slots[t]; }
  const std::vector<SparseLU::Slot>& sl() const { return E.slots[t]; }
  double slot_c_fact(int i) const { return sl()[i].c_fact; }
  double slot_crate(int i) const { return sl()[i].crate; }
  long long slot_crate_step(int i) const { return sl()[i].crate_step; }
  long long slot_crate_restart(int i) const { return sl()[i].crate_restart; }
  void slot_touch(int i, long long c) { sl()[i].last_use = c; }
  void slot_rate(int i, double cr, long long st, long long rs) { sl()[i].crate = cr; sl()[i].crate_step = st; sl()[i].crate_restart = rs; }
  void slot_drop(int i) { sl()[i].valid = false; }
  void slot_made(int i, double c, long long clock, long long js, long long ss) {
    SparseLU::Slot& q = sl()[i];
    q.c_fact = c; q.crate = 1.0; q.valid = true; q.last_use = clock; q.jac_stamp = js; q.step_stamp = ss;
  }
  void slots_invalidate(bool reset) { for (auto& q : sl()) { q.valid = false; if (reset) { q.c_fact = 0.0; q.last_use = 0; } } }
  int nearest_slot(double c, double band, long long n_restarts, long long max_age) const {
    int best = -1; double bd = 1e300;
    for (int i = 0; i < (int)sl().size(); i++) {
      const SparseLU::Slot& q = sl()[i];
      if (!q.valid || n_restarts - q.jac_stamp > max_age) continue;
      const double r = std::fabs(std::log(c / q.c_fact));
      if (r < bd && std::fabs(c / q.c_fact - 1.0) <= band) { bd = r; best = i; }
    }
    return best;
  }
  int victim_slot(long long n_restarts, long long max_age, int n_slots) const {
    for (int i = 0; i < n_slots; i++) if (!sl()[i].valid || n_restarts - sl()[i].jac_stamp > max_age) return i;
    int v = 0;
    for (int i = 1; i < n_slots; i++) if (sl()[i].last_use < sl()[v].last_use) v = i;
    return v;
  }

  // ---- vectors
  void vec(int op, double h0 = 0.0, double* out = nullptr, int order = 0, const double* p = nullptr) {
    Pending q = make(K_VEC);
    q.op.i0 = op; q.op.d0 = h0; q.op.out = out; q.op.i1 = order;
    if (p) for (int j = 0; j <= RES_MAX_ORDER; j++) q.op.p[j] = p[j];
    run(q);
  }
  void load_u0() { vec(EV_LOAD_U0, 0.0, const_cast<double*>(u0_dev)); }
  void chunk_start_from_y() { vec(EV_CS_FROM_Y); }
  void y_from_chunk_start_clipped() { vec(EV_Y_FROM_CS_CLIPPED); }
  void y_from_D0() { vec(EV_Y_FROM_D0); }
  void ytmp_from_D0() { vec(EV_YTMP_FROM_D0); }
  void ytmp_axpy(double h0) { vec(EV_YTMP_AXPY, h0); }
  double* row(long long r) { return E.sol.p + ((size_t)t * (size_t)E.cap + (size_t)r) * (size_t)E.N; }
  void save_y(long long r, double time) { E.sol_t[t][(size_t)r] = time; vec(EV_SAVE_Y, 0.0, row(r)); }
  void set_time(long long r, double time) { E.sol_t[t][(size_t)r] = time; }
  void interp(int order, const double* p, long long r) { vec(EV_INTERP, 0.0, row(r), order, p); }
  void apply_rates(long long stop) {
    Pending q = make(K_APPLY_RATES);
    if (P.rate_mode == 1) { q.op.i0 = 1; q.op.out = const_cast<double*>(table_dev) + (size_t)stop * (size_t)E.R; }
    else { q.op.i0 = 2; q.op.d0 = T_stops[stop]; }
    run(q);
  }
  void rhs(int which) { Pending q = make(K_RHS); q.op.i0 = which; run(q); }
  void rhs_y_to_f0() { rhs(0); }
  void rhs_ytmp_to_f1() { rhs(1); }
  void rhs_ytmp_to_f0() { rhs(2); }
  void eval_jac_y() { run(make(K_JAC)); }
  ResNorms norms(bool with_f1, double atol, double rtol) {
    Pending q = make(K_NORMS);
    q.op.i0 = with_f1 ? 1 : 0; q.op.d0 = atol; q.op.d1 = rtol;
    run(q);
    const BdfCtrl& c = E.ctrl_of[t];
    return ResNorms{c.scratch[0], c.scratch[1], c.scratch[2], c.scratch[3], c.nonfinite};
  }
  void init_D(bool from_ytmp, double hh) {
    pend_accept = false; pend_change = false;     // D is rebuilt: whatever was deferred on the old one is moot
    Pending q = make(K_INIT_D);
    q.op.i0 = from_ytmp ? 1 : 0; q.op.d0 = hh;
    run(q);
  }
  void predict(int order, const double* gamma, double alpha_o, double atol, double rtol) {
    // (a predictor outside a corrector attempt: the state at which a Jacobian is refreshed) - a corrector entry without
    // iterations would do; kept simple: one attempt's predictor through the corrector path with zero iterations is not
    // offered by the round, so the Jacobian refresh uses a whole corrector-less entry
    (void)gamma;
    Pending q = make(K_CORRECTOR);
    q.op.in = ResCorrIn{};
    q.op.in.order = order; q.op.in.alpha_o = alpha_o; q.op.in.atol = atol; q.op.in.rtol = rtol;
    q.op.i0 = 1;   // predictor only
    q.op.W = sl()[0].W.p; q.op.sinv = sl()[0].sinv;
    run(q);
  }
  void change_D(int ord, const double (*RU)[6]) {
    // deferred: composed with what is pending (D' = M2^T (M1^T D) = (M1 M2)^T D), applied in front of the next operation
    double M[36];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) M[i * 6 + j] = (i <= ord && j <= ord) ? RU[i][j] : (i == j ? 1.0 : 0.0);
    if (!pend_change) { std::copy(M, M + 36, pend_ru); pend_change = true; return; }
    double C[36];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double v = 0.0;
        for (int q = 0; q < 6; q++) v += pend_ru[i * 6 + q] * M[q * 6 + j];
        C[i * 6 + j] = v;
      }
    std::copy(C, C + 36, pend_ru);
  }
  void accept(int order) { pend_accept = true; pend_accept_order = order; }
  int drift_check(double max_drift) {
    Pending q = make(K_DRIFT);
    q.max_drift = max_drift;
    return run(q).dropped;
  }
  bool factor(int slot, double c, bool keep_diag) {
    return E.factor_member(t, slot, c, keep_diag);   // leaves the round: own stream, dense block to the batching server
  }
  ResAttempt corrector(const ResCorrIn& in, const double*) {
    Pending q = make(K_CORRECTOR);
    q.op.in = in;
    q.op.i0 = 0;
    q.op.W = sl()[in.slot].W.p; q.op.sinv = sl()[in.slot].sinv;
    run(q);
    if (!E.ctrl_of[t].newton_done && BLIND_ITERS < BDF_NEWTON_MAXITER) {   // not decided by the blind iterations: the rest, next round
      Pending q2 = make(K_CORRECTOR_CONT);
      q2.op = q.op;
      run(q2);
    }
    const BdfCtrl& c = E.ctrl_of[t];
    ResAttempt a{};
    a.done = c.newton_done != 0; a.converged = c.converged != 0; a.nonfinite = c.nonfinite != 0; a.any_negative = c.any_negative != 0;
    a.deep_negative = (c.any_negative & 2) != 0;
    a.n_iter = c.n_iter; a.err = c.err_norm; a.err_m = c.err_m_norm; a.err_p = c.err_p_norm; a.crate = c.crate;
    return a;
  }
};

}  // namespace

EnsembleSolver* get_ensemble(kin_network* h) {
  if (!h->ensemble) h->ensemble.reset(new EnsembleSolver(h));
  return h->ensemble.get();
}

bool ensemble_batched_supported(kin_network* h, std::string* why) {
  EnsembleSolver* E = get_ensemble(h);
  if (!E->ok && why) *why = E->why;
  return E->ok;
}

// K members of one (large) network, advanced in lockstep rounds; arguments and outputs as resident_ensemble (resident.cpp)
static void batched_ensemble_block(kin_network* h, const kin_params& p, int64_t K, const double* u0, const double* k, const double* T,
                                   const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* out_rows,
                                   double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats);

// Every member's controller runs on a host thread of its own: an ensemble of more members than KIN_ENSEMBLE_MAX_MEMBERS
// (default 128) is integrated block after block (the device is saturated long before that many members of a large network)
void batched_ensemble(kin_network* h, const kin_params& p, int64_t K, const double* u0, const double* k, const double* T,
                      const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* out_rows,
                      double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats) {
  int64_t block = 128;
  if (const char* e = getenv("KIN_ENSEMBLE_MAX_MEMBERS")) block = std::max(1, atoi(e));
  const int64_t N = h->host.N, R = h->host.R;
  const int64_t cap = make_res_grid(p).cap;
  if (K <= block) {
    batched_ensemble_block(h, p, K, u0, k, T, tstops, T_stops, k_table, n_stops, out_rows, out_t, out_u, n_saved, retcodes, stats);
    return;
  }
  // the save times are the members' common grid: they are taken from the block whose best member got furthest
  std::vector<double> tt((size_t)cap);
  std::vector<int64_t> ns((size_t)K, 0);
  int64_t best = -1;
  for (int64_t m0 = 0; m0 < K; m0 += block) {
    const int64_t n = std::min(block, K - m0);
    batched_ensemble_block(h, p, n, u0 + m0 * N, k ? k + m0 * R : nullptr, T ? T + m0 : nullptr, tstops, T_stops, k_table, n_stops, out_rows,
                           tt.data(), out_u ? out_u + m0 * cap * N : nullptr, ns.data() + m0, retcodes ? retcodes + m0 : nullptr,
                           stats ? stats + m0 : nullptr);
    const int64_t b = *std::max_element(ns.begin() + m0, ns.begin() + m0 + n);
    if (out_t && b > best) { std::copy(tt.begin(), tt.end(), out_t); best = b; }
  }
  if (n_saved) std::copy(ns.begin(), ns.end(), n_saved);
}

static void batched_ensemble_block(kin_network* h, const kin_params& p, int64_t K, const double* u0, const double* k, const double* T,
                                   const double* tstops, const double* T_stops, const double* k_table, int64_t n_stops, int64_t* out_rows,
                                   double* out_t, double* out_u, int64_t* n_saved, int32_t* retcodes, kin_stats* stats) {
  auto wall0 = std::chrono::steady_clock::now();
  EnsembleSolver& E = *get_ensemble(h);
  if (!E.ok) throw KinError(ERR_UNSUPPORTED, E.why);
  hipStream_t s = h->stream;
  const int64_t N = h->host.N, R = h->host.R;
  const ResGrid g = make_res_grid(p);
  if (out_rows) *out_rows = g.cap;
  // LU-cache slots per member: as the resident path, bounded by KIN_LU_CACHE_MB over all members
  size_t budget_mb = 32768;
  if (const char* e = getenv("KIN_LU_CACHE_MB")) budget_mb = (size_t)std::max(1, atoi(e));
  const size_t fit = std::max<size_t>(1, budget_mb * 1024 * 1024 / std::max<size_t>(1, E.lu.slot_bytes() * (size_t)K));
  int want = RES_MAX_SLOTS;
  if (const char* e = getenv("KIN_LU_CACHE_SLOTS")) want = std::max(1, atoi(e));
  const int slots = (int)std::min<size_t>((size_t)std::min(want, RES_MAX_SLOTS), fit);
  E.prepare((int)K, slots, g.cap);
  ResParams P{};
  res_fill_params(P, p, g);
  res_default_settings(P, slots);
  P.save_local = g.save_local.data();
  P.n_stops = (int32_t)n_stops;
  P.rate_mode = n_stops > 0 ? (k_table ? 1 : 2) : 0;
  P.tstops = tstops;
  DevBuf<double> d_u0;
  d_u0.upload(u0, (size_t)K * N, s);
  if (n_stops > 0 && k_table) { h->table.upload(k_table, (size_t)n_stops * R, s); h->table_rows = n_stops; }
  if (n_stops == 0) {
    for (int64_t t = 0; t < K; t++) {
      if (k) KIN_HIP(hipMemcpyAsync(E.reps[t].k, k + t * R, (size_t)R * sizeof(double), hipMemcpyHostToDevice, s));
      else if (T) launch_arrhenius(R, h->Ea.p, h->A.p, h->has_kmax, h->k_max, h->t_mult, T[t], E.reps[t].k, s);
      else KIN_HIP(hipMemcpyAsync(E.reps[t].k, h->k.p, (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
  }
  KIN_HIP(hipStreamSynchronize(s));
  std::vector<ResResult> res((size_t)K);
  std::vector<std::string> errs((size_t)K);
  std::vector<std::thread> th;
  E.start_gj_server();
  th.reserve((size_t)K);
  std::string spawn_err;
  for (int64_t t = 0; t < K; t++) {
    try {
      th.emplace_back([&, t] {
        try {
          (void)hipSetDevice(h->device);
          MemberBackend b(E, (int)t, P, d_u0.p + (size_t)t * N, n_stops > 0 && k_table ? h->table.p : nullptr, T_stops);
          ResidentBdf<MemberBackend> ctl(b, P);
          res[t] = ctl.run();
        } catch (const std::exception& e) {
          errs[t] = e.what();
          res[t] = ResResult{};
          res[t].retcode = RES_RET_UNSTABLE;
        }
        E.member_done();
      });
    } catch (const std::exception& e) {
      // the host refused another thread: the members that never started leave the rounds (or the started ones would wait for them
      // for ever), the started ones are joined (a joinable std::thread that goes out of scope terminates the process)
      spawn_err = e.what();
      for (int64_t q = t; q < K; q++) { res[q] = ResResult{}; res[q].retcode = RES_RET_UNSTABLE; E.member_done(); }
      break;
    }
  }
  for (auto& x : th) x.join();
  E.stop_gj_server();
  if (!spawn_err.empty()) throw KinError(ERR_DEVICE, "ensemble: could not start a member thread: " + spawn_err);
  for (auto& e : errs) if (!e.empty()) throw KinError(ERR_DEVICE, "ensemble member failed: " + e);
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
  if (out_u) E.sol.download(out_u, (size_t)K * (size_t)g.cap * N, s);
  KIN_HIP(hipStreamSynchronize(s));
  int64_t best = 0;
  for (int64_t t = 1; t < K; t++) if (res[t].n_saved > res[best].n_saved) best = t;
  if (out_t) std::copy(E.sol_t[best].begin(), E.sol_t[best].begin() + g.cap, out_t);
  if (getenv("KIN_TIMING")) {
    fprintf(stderr, "[ensemble] %lld members, %lld rounds, wall %.4f s: inside rounds %.4f s (enqueue %.4f, waiting for the device %.4f), "
            "between rounds %.4f s\n", (long long)K, (long long)E.n_rounds, wall, E.t_round, E.t_enqueue, E.t_sync, wall - E.t_round);
    fprintf(stderr, "[ensemble] round hand-overs: %lld through pinned memory, %lld by copy + stream synchronisation\n", (long long)E.n_fast, (long long)E.n_slow);
    fprintf(stderr, "[ensemble] dense inverses: %lld in %lld batched chains (%.2f per chain)\n", (long long)E.g_matrices, (long long)E.g_batches,
            E.g_batches ? (double)E.g_matrices / (double)E.g_batches : 0.0);
    fprintf(stderr, "[ensemble] operations: vec %lld, rates %lld, rhs %lld, jac %lld, norms %lld, init_D %lld, drift %lld, factor %lld, corrector %lld (+%lld continued)\n",
            (long long)E.n_ops[K_VEC], (long long)E.n_ops[K_APPLY_RATES], (long long)E.n_ops[K_RHS], (long long)E.n_ops[K_JAC], (long long)E.n_ops[K_NORMS],
            (long long)E.n_ops[K_INIT_D], (long long)E.n_ops[K_DRIFT], (long long)E.n_ops[K_FACTOR], (long long)E.n_ops[K_CORRECTOR], (long long)E.n_ops[K_CORRECTOR_CONT]);
  }
  for (int64_t t = 0; t < K; t++) {
    if (n_saved) n_saved[t] = std::min<int64_t>(res[t].n_saved, g.cap);
    if (retcodes) retcodes[t] = res[t].retcode;
    if (stats) {
      kin_stats& st = stats[t];
      st = kin_stats{};
      const ResStats& q = res[t].st;
      st.n_steps = q.n_steps; st.n_rejected = q.n_rejected; st.n_rhs = q.n_rhs; st.n_jac = q.n_jac; st.n_factor = q.n_factor;
      st.n_linsolve = q.n_linsolve; st.n_newton_fail = q.n_newton_fail; st.n_chunks = q.n_chunks; st.n_restarts = q.n_restarts;
      st.n_retries = q.n_retries; st.final_abstol = res[t].final_abstol; st.final_reltol = res[t].final_reltol; st.wall_seconds = wall;
      st.lu_dense_dim = E.lu.m; st.lu_sparse_rows = E.lu.ns; st.lu_rounds = E.lu.nrounds;
      st.lu_nnz = 2 * E.lu.nnzU + E.lu.ns + (int64_t)E.lu.m * E.lu.m;
      st.n_lu_reused = q.n_lu_reused; st.lu_slots = slots; st.n_bad_pivot = q.n_bad_pivot; st.n_lu_dropped = q.n_lu_dropped;
    }
  }
}

}  // namespace kin
