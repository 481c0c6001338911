// Tiled batched RHS sweep (gfx950): B states in the library order of tiled.hpp, ONE pass over each state's rate
// constants, every random access in LDS, every global access a coalesced stream:
//   per state:   hubs' u            HBM -> LDS   (coalesced, prefetched during the previous state's last segment)
//   per segment: window's u         HBM -> LDS   (coalesced, prefetched during the previous segment)
//                records + k        streamed once, two batches of 4 record rows in flight per thread
//                window's du        LDS -> HBM   (coalesced)
//   per state:   hubs' du           LDS -> HBM
// HBM traffic = the algorithmic one (k[b] + u[b] in, du[b] out; records and tables are shared by all states and stay
// in L2 / Infinity Cache). A record is four 14-bit LDS labels with fixed roles (fields 0, 1 = reactant instances of the
// forward reaction, fields 2, 3 = its product instances; net = kf u0 u1 - kr u2 u3, du[0,1] -= net, du[2,3] += net)
// plus three flag bits that are equal for the 64 records a wavefront processes together: bit 56 = some record of the
// group has a second reactant, bit 57 = ... a second product, bit 58 = nothing to do. All four fields are always read
// (an unused field sits on a per-lane dummy entry with u = 1, so there is no branch between the LDS reads); a group
// without second reactants / products skips that field's ds_add_f64 as a whole wavefront (3.1 instead of 4 atomics per
// record on the synthetic CRNs).
//
// TMODE: no k stream at all - the rate constants are formed inside the sweep from the state's temperature and the
// records' Arrhenius parameters (SURVEY 8(d) M1'; the reference's continuous-rate path inlines k(T(t)) into every
// species ODE the same way, src/solving/methods.jl:389-419, calculator.jl:223-226).
#include "tiled_kernels.hpp"

#include "exp_tab.hpp"

#ifndef KIN_TILED_SUB
#define KIN_TILED_SUB 2
#endif

namespace kin {

namespace {

// rows per batch: 4 with a k stream (two batches = 8 rows of 16-byte loads in flight per thread cover the HBM latency);
// 2 when the rate constants are computed (the parameters come from L2 and take twice the registers)
// (a state that fits LDS whole carries 10 staged doubles per thread instead of 5: two rows per batch there as well)
template <bool TMODE, int NB_> struct Batch;
template <int NB_> struct Batch<false, NB_> { static constexpr int NB = NB_; double2 k[NB]; uint2 w[NB]; };
template <int NB_> struct Batch<true, NB_> { static constexpr int NB = NB_; double4 p[NB]; uint2 w[NB]; };

struct Labels { uint32_t l0, l1, l2, l3; };
__device__ __forceinline__ Labels decode(uint2 w) {
  return Labels{w.x & 0x3fffu, (w.x >> 14) & 0x3fffu, (w.x >> 28) | ((w.y & 0x3ffu) << 4), (w.y >> 10) & 0x3fffu};
}

}  // namespace

// UN = doubles per thread of the staged-in set (the hubs, or one window) that travel HBM -> registers ahead of their
// use: 5 when hubs and windows have at most 5 BS entries each (the windowed layouts), else 10 (a state that fits LDS whole)
// SINGLES: the layout has records with one rate-constant slot at the end of its segments (tiled.hpp); without them the
// record at position p reads k_lib[2 p], k_lib[2 p + 1] and none of the extra address arithmetic is compiled in
template <int BS, bool TMODE, int UN, bool SINGLES>
__global__ __launch_bounds__(BS) void tiled_sweep_kernel(TiledView v, int B, const double* __restrict__ u,
                                                         const double* __restrict__ k_lib, const double* __restrict__ Tb,
                                                         double* __restrict__ du) {
  extern __shared__ double lds[];
  double* du_s = lds;
  double* u_s = lds + v.E;
  const double* tab_s = lds + 2 * v.E;
  const int tid = threadIdx.x;
  const int N = v.N, h = v.h, wbase = v.wbase, T = v.T;
  const size_t KL = (size_t)v.KL;

  for (int i = tid; i < v.E; i += BS) du_s[i] = 0.0;
  if (tid < TILED_DUMMY) u_s[h + tid] = 1.0;
  if (TMODE) for (int i = tid; i < TILED_EXP_TAB; i += BS) lds[2 * v.E + i] = kExp2Tab[i * (512 / TILED_EXP_TAB)];
  const int csrc = tid < v.n_copy ? v.copy_src[min(tid, max(v.n_copy - 1, 0))] : -1;
  // padding record of this lane: every field on the lane's dummy entry, "nothing to do" set
  const uint64_t dl = (uint64_t)(h + (tid & 63));
  const uint64_t ew = dl | (dl << 14) | (dl << 28) | (dl << 42) | (4ull << 56);
  const uint2 EMPTY = {(uint32_t)ew, (uint32_t)(ew >> 32)};

  // Every global load below is UNCONDITIONAL (indices clamped into valid memory, the value replaced afterwards): a load
  // behind a branch makes the compiler wait for vmcnt(0) wherever it cannot count the loads in flight, which would drain
  // the two-batch queue at every use (seen in the first version of this kernel: 0.78 -> 0.64 ms at C5).
  const int Bm1 = B - 1;
  // ---- the staged-in set of the next phase travels HBM -> registers while the current phase computes
  double un[UN];
  // the hubs of state bb
  auto load_hubs = [&](int bb) {
    const double* ub = u + (size_t)min(bb, Bm1) * N;
#pragma unroll
    for (int x = 0; x < UN; x++) un[x] = ub[min(tid + x * BS, h - 1)];
  };
  auto store_hubs = [&]() {
#pragma unroll
    for (int x = 0; x < UN; x++) {
      const int e = tid + x * BS;
      if (e < h) u_s[e] = un[x];
    }
  };
  auto load_window = [&](int bb, int s) {
    const double* ub = u + (size_t)bb * N + v.win_off[s];
    const int cm1 = v.win_cnt[s] - 1;      // (windows are never empty when there are any)
#pragma unroll
    for (int x = 0; x < UN; x++) un[x] = ub[min(tid + x * BS, cm1)];
  };
  auto store_window = [&](int s) {
    const int c = v.win_cnt[s];
#pragma unroll
    for (int x = 0; x < UN; x++) {
      const int e = tid + x * BS;
      if (e < c) u_s[wbase + e] = un[x];
    }
  };

  // ---- record / rate-constant queue: two batches of NB record rows in flight per thread. The producer runs 2 NB
  // iteration rows ahead of the consumer, across segment and state boundaries. The segment descriptors are part of the
  // kernel arguments (scalar loads from the kernarg segment); `pa` = the producer's segment, `pn` = the one after, re-read
  // only at the consumer's segment starts, right behind a barrier (a scalar load's wait also drains the LDS queue). Every
  // segment has >= TILED_MIN_ROWS >= 2 NB iteration rows, a whole number of batches (tiled.hpp: row_quantum), and the
  // producer changes segment lazily (at the first request beyond the end), so it does so at most once in between.
  constexpr int NB = (TMODE || UN > 5) ? 2 : 4;
  using BatchT = Batch<TMODE, NB>;
  static_assert(TILED_MIN_ROWS >= 2 * NB && (UN > 5 ? 2 : 4) % NB == 0, "segments are padded to whole batches, two at least");
  int pb = blockIdx.x, ps = 0, pi = 0;   // state, segment, iteration row inside the segment of the next request
  int4 pa = v.seginfo[0], pn = v.seginfo[T > 1 ? 1 : 0];   // x = first record, y = records, z = iteration rows, w = two-slot records
  int pak = SINGLES ? v.segk[0] : 0, pnk = SINGLES ? v.segk[T > 1 ? 1 : 0] : 0;   // slot of the segment's first record
  asm volatile("" : "+s"(pa.x), "+s"(pa.y), "+s"(pa.z), "+s"(pn.x), "+s"(pn.y), "+s"(pn.z));
  if constexpr (SINGLES) asm volatile("" : "+s"(pa.w), "+s"(pn.w), "+s"(pak), "+s"(pnk));
  auto load_batch = [&](BatchT& G) {
    if (pi >= pa.z) {
      pi = 0; pa = pn; pak = pnk;
      if (++ps == T) { ps = 0; pb += gridDim.x; }
    }
    const bool live = pb < B;
    const double* kb = TMODE ? nullptr : k_lib + (size_t)min(pb, Bm1) * KL;
#pragma unroll
    for (int x = 0; x < NB; x++) {
      const int off = (pi + x) * BS + tid;
      const bool ok = live && off < pa.y;
      const int p = ok ? pa.x + off : 0;
      const uint2 w = v.rec[p];
      G.w[x].x = ok ? w.x : EMPTY.x;
      G.w[x].y = ok ? w.y : EMPTY.y;
      if constexpr (TMODE) G.p[x] = v.par[p];
      else if constexpr (!SINGLES) G.k[x] = *reinterpret_cast<const double2*>(kb + 2 * (size_t)p);
      else {
        // a wavefront's 64 records are all of one kind (n2 is a multiple of 64 wherever one-slot records follow): two
        // neighbouring one-slot records share a 16-byte load, the same instruction as for the two-slot records
        const bool one = (off & ~63) >= pa.w;
        const size_t e = !ok ? 0 : (one ? (size_t)(pak + pa.w) + (size_t)(off & ~1) : (size_t)pak + 2 * (size_t)off);
        const double2 kk = *reinterpret_cast<const double2*>(kb + e);
        // (selects on the 32-bit halves: a select between the two components of one double2 value becomes an indexed
        // element access and sends the whole queue to scratch memory)
        const bool odd = one && (off & 1);
        const int lo = odd ? __double2loint(kk.y) : __double2loint(kk.x), hi = odd ? __double2hiint(kk.y) : __double2hiint(kk.x);
        const int rlo = one ? 0 : __double2loint(kk.y), rhi = one ? 0 : __double2hiint(kk.y);
        G.k[x].x = __hiloint2double(hi, lo);
        G.k[x].y = __hiloint2double(rhi, rlo);
      }
    }
    pi += NB;
  };

  double RT = 1.0, inv_RT = 1.0;
#if defined(KIN_TILED_PROBE)
  double probe_acc = 0.0;
#endif
  // Rows of a batch are processed two at a time: the pair's eight LDS reads in flight together, both net rates formed,
  // then the pair's atomics back to back - nothing waits between them, and the next pair's reads queue up behind them
  // (the LDS executes a wave's operations in order). An all-padding row has EMPTY records: rate constants times
  // dummy entries, added to dummy entries; its wavefronts skip the atomics.
  auto consume = [&](const BatchT& G) {
    constexpr int SUB = KIN_TILED_SUB < NB ? KIN_TILED_SUB : NB;
#pragma unroll
    for (int x0 = 0; x0 < NB; x0 += SUB) {
      uint32_t fl[SUB];
      double net[SUB];
#pragma unroll
      for (int y = 0; y < SUB; y++) {
        // all four fields are read whatever the flags say (an unused field sits on the lane's dummy entry, u = 1): no
        // branch between the LDS reads
        const int x = x0 + y;
        fl[y] = __builtin_amdgcn_readfirstlane(G.w[x].y >> 24);
        const Labels L = decode(G.w[x]);
#if defined(KIN_TILED_SKIPREAD)
        double u1v = 1.0, u3v = 1.0;
        const double u0v = u_s[L.l0];
        if (fl[y] & 1u) u1v = u_s[L.l1];
        const double u2v = u_s[L.l2];
        if (fl[y] & 2u) u3v = u_s[L.l3];
        const double uf = u0v * u1v, ur = u2v * u3v;
#elif defined(KIN_TILED_PROBE) && (KIN_TILED_PROBE == 1 || KIN_TILED_PROBE == 3)   // timing only (wrong results): no LDS operand reads
        const double uf = 1.0 + (double)L.l0, ur = 1.0 + (double)L.l2;
#else
        const double uf = u_s[L.l0] * u_s[L.l1];
        const double ur = u_s[L.l2] * u_s[L.l3];
#endif
        double kf, kr;
        if constexpr (TMODE) {
          kf = arrhenius_fast_t<TILED_EXP_TAB>(G.p[x].x, G.p[x].y, G.p[x].y, RT, inv_RT, v.has_kmax, v.inv_kmax, tab_s);
          kr = arrhenius_fast_t<TILED_EXP_TAB>(G.p[x].z, G.p[x].w, G.p[x].w, RT, inv_RT, v.has_kmax, v.inv_kmax, tab_s);
        } else {
          kf = G.k[x].x; kr = G.k[x].y;
        }
        net[y] = kf * uf - kr * ur;
      }
#pragma unroll
      for (int y = 0; y < SUB; y++) {
        if (fl[y] & 4u) continue;
        const Labels L = decode(G.w[x0 + y]);
#if defined(KIN_TILED_PROBE)   // timing only (wrong results): no LDS atomics, the net rate kept alive in a register
        probe_acc += net[y];
        continue;
#endif
        __hip_atomic_fetch_add(du_s + L.l0, -net[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (fl[y] & 1u) __hip_atomic_fetch_add(du_s + L.l1, -net[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(du_s + L.l2, net[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (fl[y] & 2u) __hip_atomic_fetch_add(du_s + L.l3, net[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };

  int b = blockIdx.x;
  load_hubs(b);
  const int csrc_c = max(csrc, 0);
  double ucn = u[(size_t)min(b, Bm1) * N + csrc_c];
  double Tn = TMODE ? Tb[min(b, Bm1)] : 1.0;
  BatchT G0, G1;
  load_batch(G0);
  load_batch(G1);

  for (; b < B; b += gridDim.x) {
    double* dub = du + (size_t)b * N;
    const int bn = b + gridDim.x;
    if (TMODE) { RT = 8.314462618 * Tn; inv_RT = 1.0 / RT; }
#if defined(KIN_TILED_PROBE) && KIN_TILED_PROBE == 3   // timing only: the record / k queue alone, no staging, barriers or write-out
    for (int s = 0; s < T; s++) {
      pn = v.seginfo[ps + 1 == T ? 0 : ps + 1];
      if constexpr (SINGLES) pnk = v.segk[ps + 1 == T ? 0 : ps + 1];
      int rows = v.seginfo[s].z;
      asm volatile("" : "+s"(pn.x), "+s"(pn.y), "+s"(pn.z), "+s"(rows));
      int q = 0;
      for (; q + 2 * NB <= rows; q += 2 * NB) { consume(G0); load_batch(G0); consume(G1); load_batch(G1); }
      if (q < rows) {
        consume(G0);
#pragma unroll
        for (int x = 0; x < NB; x++) {
          G0.w[x].x = G1.w[x].x; G0.w[x].y = G1.w[x].y;
          if constexpr (TMODE) { G0.p[x].x = G1.p[x].x; G0.p[x].y = G1.p[x].y; G0.p[x].z = G1.p[x].z; G0.p[x].w = G1.p[x].w; }
          else { G0.k[x].x = G1.k[x].x; G0.k[x].y = G1.k[x].y; }
        }
        load_batch(G1);
      }
    }
    if (b >= 0) continue;
#endif
    store_hubs();
    if (csrc >= 0) u_s[h + TILED_DUMMY + tid] = ucn;
    // the state's first window is not prefetched (the registers carry the hubs across the state boundary): one exposed
    // global -> LDS copy per state
    if (v.win_cnt_max > 0) { load_window(b, 0); store_window(0); }
    __syncthreads();
    ucn = u[(size_t)min(bn, Bm1) * N + csrc_c];
    if (TMODE) Tn = Tb[min(bn, Bm1)];
    for (int s = 0; s < T; s++) {
      const bool last = s == T - 1;
      if (last) load_hubs(bn); else load_window(b, s + 1);
      pn = v.seginfo[ps + 1 == T ? 0 : ps + 1];
      if constexpr (SINGLES) pnk = v.segk[ps + 1 == T ? 0 : ps + 1];
      int rows = v.seginfo[s].z;
      // the scalar loads above are waited for HERE, behind the barrier: left to the compiler, the wait (lgkmcnt(0), which
      // also drains the wave's LDS queue) lands on the join inside the record loop, once per batch
      asm volatile("" : "+s"(pn.x), "+s"(pn.y), "+s"(pn.z), "+s"(rows));
      if constexpr (SINGLES) asm volatile("" : "+s"(pn.w), "+s"(pnk));
      int q = 0;
      for (; q + 2 * NB <= rows; q += 2 * NB) {
        consume(G0);
        load_batch(G0);
        consume(G1);
        load_batch(G1);
      }
      if (q < rows) {   // an odd number of batches: the queue keeps its order (G0 = the older batch) by a register move
        consume(G0);
#pragma unroll
        for (int x = 0; x < NB; x++) {   // (component by component: whole-vector copies of register arrays end up in scratch)
          G0.w[x].x = G1.w[x].x; G0.w[x].y = G1.w[x].y;
          if constexpr (TMODE) { G0.p[x].x = G1.p[x].x; G0.p[x].y = G1.p[x].y; G0.p[x].z = G1.p[x].z; G0.p[x].w = G1.p[x].w; }
          else { G0.k[x].x = G1.k[x].x; G0.k[x].y = G1.k[x].y; }
        }
        load_batch(G1);
      }
      __syncthreads();
      if (last && v.n_copy > 0) {   // fold the split accumulators back into their species
        if (csrc >= 0) {
          const double a = du_s[h + TILED_DUMMY + tid];
          du_s[h + TILED_DUMMY + tid] = 0.0;
          __hip_atomic_fetch_add(du_s + csrc, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
      }
      // window s out (and re-zeroed); on the state's last segment the hubs too
      {
        const int c = v.win_cnt[s], o = v.win_off[s];
        for (int i = tid; i < c; i += BS) { dub[o + i] = du_s[wbase + i]; du_s[wbase + i] = 0.0; }
        if (last) for (int i = tid; i < h; i += BS) { dub[i] = du_s[i]; du_s[i] = 0.0; }
      }
      if (!last) {
        store_window(s + 1);   // u_s of the window region is dead since the barrier above
        __syncthreads();
      }
      // (after the last segment the next state's head is stored at the top of the loop, before its barrier)
    }
  }
#if defined(KIN_TILED_PROBE)
  if (probe_acc == 12345.678) du[0] = probe_acc;
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// layout helpers: every kernel below is a coalesced write of the destination with a gather on the source side
// ------------------------------------------------------------------------------------------------------------------
// dst[b][j] = src[b][map[j]]  (map[j] < 0: 0.0)
__global__ __launch_bounds__(256) void gather_rows_kernel(int64_t n_dst, int64_t n_src, int B, const int32_t* __restrict__ map,
                                                          const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_dst) return;
  const int32_t m = map[j];
  for (int b = blockIdx.y; b < B; b += gridDim.y) dst[(size_t)b * n_dst + j] = m >= 0 ? src[(size_t)b * n_src + m] : 0.0;
}

// The species permutation of the layout conversion, staged through LDS (tiled.hpp). One workgroup = one piece of one state:
//   TO_LIB   the piece of the caller's row comes in as one coalesced stream, goes into LDS, and leaves as the runs of the
//            library row its species belong to (one run per hub / window range: hundreds of consecutive doubles each);
//   else     the runs come in, the piece goes out.
// 8 N bytes in and 8 N bytes out per state, nothing fetched twice: the plain gather above reads (or writes) 8-byte entries
// scattered over the whole row, a 64-byte request each once the row has left the cache (C5, B = 1024: 0.35 / 0.21 ms for the
// two directions; DESIGN 3.1b has this kernel's numbers). All loads of a thread are issued before its first LDS access:
// 32 elements per thread = 64 kB in flight per workgroup.
// VEC2: the rows are 16-byte aligned (N even, aligned bases) - the piece side moves 16 bytes per lane.
template <bool TO_LIB, bool VEC2>
__global__ __launch_bounds__(256) void permute_staged_kernel(int N, const int32_t* __restrict__ stage_lib, const int32_t* __restrict__ stage_off,
                                                             const double* __restrict__ src, double* __restrict__ dst) {
  __shared__ double2 buf2[TILED_PIECE / 2];
  double* buf = (double*)buf2;
  constexpr int PT = TILED_PIECE / 256;
  const int p0 = blockIdx.x * TILED_PIECE;
  const int cnt = min(N - p0, TILED_PIECE);
  const double* s = src + (size_t)blockIdx.y * N;
  double* d = dst + (size_t)blockIdx.y * N;
  int32_t jl[PT], of[PT];
#pragma unroll
  for (int x = 0; x < PT; x++) {
    const int q = threadIdx.x + 256 * x;
    const bool ok = q < cnt;
    jl[x] = ok ? stage_lib[p0 + q] : -1;
    of[x] = ok ? stage_off[p0 + q] : 0;
  }
  if (TO_LIB) {
    if (VEC2) {   // (cnt is even whenever N is: every piece but the last is whole)
      double2 v2[PT / 2];
#pragma unroll
      for (int x = 0; x < PT / 2; x++) { const int q = threadIdx.x + 256 * x; v2[x] = 2 * q < cnt ? ((const double2*)(s + p0))[q] : make_double2(0.0, 0.0); }
#pragma unroll
      for (int x = 0; x < PT / 2; x++) { const int q = threadIdx.x + 256 * x; if (2 * q < cnt) buf2[q] = v2[x]; }
    } else {
      double v[PT];
#pragma unroll
      for (int x = 0; x < PT; x++) { const int q = threadIdx.x + 256 * x; v[x] = q < cnt ? s[p0 + q] : 0.0; }
#pragma unroll
      for (int x = 0; x < PT; x++) { const int q = threadIdx.x + 256 * x; if (q < cnt) buf[q] = v[x]; }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < PT; x++) if (jl[x] >= 0) d[jl[x]] = buf[of[x]];
  } else {
    double v[PT];
#pragma unroll
    for (int x = 0; x < PT; x++) v[x] = jl[x] >= 0 ? s[jl[x]] : 0.0;
#pragma unroll
    for (int x = 0; x < PT; x++) if (jl[x] >= 0) buf[of[x]] = v[x];
    __syncthreads();
    if (VEC2) {
#pragma unroll
      for (int x = 0; x < PT / 2; x++) { const int q = threadIdx.x + 256 * x; if (2 * q < cnt) ((double2*)(d + p0))[q] = buf2[q]; }
    } else {
#pragma unroll
      for (int x = 0; x < PT; x++) { const int q = threadIdx.x + 256 * x; if (q < cnt) d[p0 + q] = buf[q]; }
    }
  }
}

// k_lib[b][2 p], k_lib[b][2 p + 1] = k[b][kf[p]], k[b][kr[p]] for a layout whose records all have two slots (k_lib row = 2 P):
// one thread per record. Where the caller keeps a reaction and its reverse next to each other at an even position - the
// usual case - the pair travels as ONE 16-byte load (swapped when the library made the reverse the record's forward
// reaction); otherwise two 8-byte gathers. The record order is the caller's reaction order permuted inside chunks of 64
// records (tiled.cpp), so a wavefront's loads fall into one kB: the slot-wise gather above issues two instructions of 64
// scattered 8-byte lanes for what is one instruction of 64 16-byte lanes here (with the wider scramble of a 256-record
// reservoir: C3 1.13 -> 1.06 ms, C5 1.63 -> 1.58 ms; with the chunked default 0.82 / 1.45 ms).
__global__ __launch_bounds__(256) void rates_to_lib_pairs_kernel(int P, int64_t R, int B, const int32_t* __restrict__ kf,
                                                                 const int32_t* __restrict__ kr, const double* __restrict__ src,
                                                                 double* __restrict__ dst) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int f = kf[p], r = kr[p];
  const bool straight = r == f + 1 && (f & 1) == 0, flipped = f == r + 1 && (r & 1) == 0;
  const size_t lo = (size_t)(straight ? f : (flipped ? r : 0));
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const double* sb = src + (size_t)b * R;
    double2 v;
    if (straight || flipped) {
      const double2 w = *reinterpret_cast<const double2*>(sb + lo);
      v.x = flipped ? w.y : w.x;
      v.y = flipped ? w.x : w.y;
    } else {
      v.x = sb[f];
      v.y = r >= 0 ? sb[r] : 0.0;
    }
    *reinterpret_cast<double2*>(dst + ((size_t)b * P + p) * 2) = v;
  }
}

// par[p] = (Ea_f, X_f, Ea_r, X_r), X = 1 / (A N_A t_mult) when the calculator caps its rate constants (the sweep and the
// table kernel evaluate 1 / (1/k_max + X e^q)), else A N_A t_mult. A missing reverse: a constant that makes k = 0.
__global__ __launch_bounds__(256) void tiled_params_kernel(int P, const int32_t* __restrict__ kf, const int32_t* __restrict__ kr,
                                                           const double* __restrict__ Ea, const double* __restrict__ A, int has_kmax,
                                                           double t_mult, double4* __restrict__ par) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int f = kf[p], r = kr[p];
  const double cf = A[f] * 6.02214076e23 * t_mult;
  const double none = has_kmax ? __builtin_inf() : 0.0;
  double4 o;
  o.x = Ea[f]; o.y = has_kmax ? 1.0 / cf : cf;
  o.z = r >= 0 ? Ea[r] : 0.0;
  o.w = r >= 0 ? (has_kmax ? 1.0 / (A[r] * 6.02214076e23 * t_mult) : A[r] * 6.02214076e23 * t_mult) : none;
  par[p] = o;
}

// rate table in library order: table[s][slot], table[s][slot + 1] = forward / reverse rate constant of record p at T[s]
// (kslot[p] >= 0: a two-slot record; ~slot: a one-slot record, forward constant only; KL = slots per row)
__global__ __launch_bounds__(256) void rate_table_lib_kernel(int P, int64_t KL, int n_stops, const double4* __restrict__ par,
                                                             const int32_t* __restrict__ kslot, TiledPadSlots pads, int has_kmax, double inv_kmax,
                                                             const double* __restrict__ T, double* __restrict__ table) {
  constexpr int ROWS = 32;
  __shared__ double rt_s[ROWS], irt_s[ROWS];
  __shared__ double tab_s[512];
  tab_s[threadIdx.x] = kExp2Tab[threadIdx.x];
  tab_s[threadIdx.x + 256] = kExp2Tab[threadIdx.x + 256];
  const int s0 = blockIdx.y * ROWS, s1 = min(n_stops, s0 + ROWS);
  if ((int)threadIdx.x < s1 - s0) {
    const double RT = 8.314462618 * T[s0 + threadIdx.x];
    rt_s[threadIdx.x] = RT;
    irt_s[threadIdx.x] = 1.0 / RT;
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < pads.n) for (int s = s0; s < s1; s++) table[(size_t)s * KL + pads.slot[p]] = 0.0;
  if (p >= P) return;
  const double4 q = par[p];
  const int32_t ks = kslot[p];
  const bool two = ks >= 0;
  const size_t slot = (size_t)(two ? ks : ~ks);
  for (int s = s0; s < s1; s++) {
    const double RT = rt_s[s - s0], inv_RT = irt_s[s - s0];
    const double kf = arrhenius_fast(q.x, q.y, q.y, RT, inv_RT, has_kmax, inv_kmax, tab_s);
    if (two) {
      const double kr = arrhenius_fast(q.z, q.w, q.w, RT, inv_RT, has_kmax, inv_kmax, tab_s);
      *reinterpret_cast<double2*>(table + (size_t)s * KL + slot) = make_double2(kf, kr);
    } else {
      table[(size_t)s * KL + slot] = kf;
    }
  }
}

void launch_gather_rows(int64_t n_dst, int64_t n_src, int64_t B, const int32_t* map, const double* src, double* dst, hipStream_t s) {
  if (n_dst == 0 || B == 0) return;
  dim3 grid((unsigned)ceil_div(n_dst, 256), (unsigned)std::min<int64_t>(B, 1024));
  hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, s, n_dst, n_src, (int)B, map, src, dst);
  KIN_HIP(hipGetLastError());
}

void launch_permute_staged(int64_t N, int64_t B, bool to_lib, const int32_t* stage_lib, const int32_t* stage_off, const double* src,
                           double* dst, hipStream_t s) {
  if (N == 0 || B == 0) return;
  // (blockIdx.y carries the state: 65 535 states per launch)
  for (int64_t b0 = 0; b0 < B; b0 += 65535) {
    const int64_t nb = std::min<int64_t>(65535, B - b0);
    dim3 grid((unsigned)ceil_div(N, (int64_t)TILED_PIECE), (unsigned)nb);
    const double* sp = src + (size_t)b0 * N;
    double* dp = dst + (size_t)b0 * N;
    const bool vec2 = N % 2 == 0 && ((((uintptr_t)sp) | ((uintptr_t)dp)) & 15) == 0;
    if (to_lib && vec2) hipLaunchKernelGGL((permute_staged_kernel<true, true>), grid, dim3(256), 0, s, (int)N, stage_lib, stage_off, sp, dp);
    else if (to_lib) hipLaunchKernelGGL((permute_staged_kernel<true, false>), grid, dim3(256), 0, s, (int)N, stage_lib, stage_off, sp, dp);
    else if (vec2) hipLaunchKernelGGL((permute_staged_kernel<false, true>), grid, dim3(256), 0, s, (int)N, stage_lib, stage_off, sp, dp);
    else hipLaunchKernelGGL((permute_staged_kernel<false, false>), grid, dim3(256), 0, s, (int)N, stage_lib, stage_off, sp, dp);
  }
  KIN_HIP(hipGetLastError());
}

void launch_rates_to_lib_pairs(int P, int64_t R, int64_t B, const int32_t* kf, const int32_t* kr, const double* src, double* dst,
                               hipStream_t s) {
  if (P == 0 || B == 0) return;
  dim3 grid((unsigned)ceil_div(P, 256), (unsigned)std::min<int64_t>(B, 1024));
  hipLaunchKernelGGL(rates_to_lib_pairs_kernel, grid, dim3(256), 0, s, P, R, (int)B, kf, kr, src, dst);
  KIN_HIP(hipGetLastError());
}

void launch_tiled_params(int P, const int32_t* kf, const int32_t* kr, const double* Ea, const double* A, int has_kmax,
                         double t_mult, void* par, hipStream_t s) {
  if (P == 0) return;
  hipLaunchKernelGGL(tiled_params_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, s, P, kf, kr, Ea, A, has_kmax, t_mult,
                     (double4*)par);
  KIN_HIP(hipGetLastError());
}

void launch_rate_table_lib(int P, int64_t KL, int64_t n_stops, const void* par, const int32_t* kslot, const TiledPadSlots& pads,
                           int has_kmax, double k_max, const double* T, double* table, hipStream_t s) {
  if (P == 0 || n_stops == 0) return;
  dim3 grid((unsigned)ceil_div(P, 256), (unsigned)ceil_div(n_stops, 32));
  hipLaunchKernelGGL(rate_table_lib_kernel, grid, dim3(256), 0, s, P, KL, (int)n_stops, (const double4*)par, kslot, pads, has_kmax, 1.0 / k_max, T, table);
  KIN_HIP(hipGetLastError());
}

template <int BS, bool TMODE, int UN, bool SINGLES>
static void launch_tiled_t(const TiledView& v, int grid, size_t smem, int B, const double* u, const double* k_lib, const double* Tb,
                           double* du, hipStream_t s) {
  // per launch, not cached: the attribute belongs to the (function, device) pair and costs ~1 us
  KIN_HIP(hipFuncSetAttribute((const void*)tiled_sweep_kernel<BS, TMODE, UN, SINGLES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((tiled_sweep_kernel<BS, TMODE, UN, SINGLES>), dim3(grid), dim3(BS), smem, s, v, B, u, k_lib, Tb, du);
}

void launch_tiled_sweep(const TiledView& v, int bs, int n_cu, int64_t B, const double* u, const double* k_lib, const double* Tb,
                        double* du, hipStream_t s) {
  if (B == 0) return;
  const bool tmode = k_lib == nullptr;
  const size_t smem = ((size_t)2 * v.E + (tmode ? TILED_EXP_TAB : 0)) * 8;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / bs, (160 * 1024) / smem));
  const int grid = (int)std::min<int64_t>(B, (int64_t)n_cu * per_cu);
  const bool un5 = std::max(v.h, v.win_cnt_max) <= 5 * bs;
  const bool singles = !tmode && v.has_singles;   // (the temperature form reads no rate constants: one instantiation)
#define KIN_TILED_GO(BSZ)                                                                                  \
  do {                                                                                                     \
    if (tmode) { if (un5) launch_tiled_t<BSZ, true, 5, false>(v, grid, smem, (int)B, u, k_lib, Tb, du, s);           \
                 else launch_tiled_t<BSZ, true, 10, false>(v, grid, smem, (int)B, u, k_lib, Tb, du, s); }            \
    else if (singles) { if (un5) launch_tiled_t<BSZ, false, 5, true>(v, grid, smem, (int)B, u, k_lib, Tb, du, s);    \
                        else launch_tiled_t<BSZ, false, 10, true>(v, grid, smem, (int)B, u, k_lib, Tb, du, s); }     \
    else { if (un5) launch_tiled_t<BSZ, false, 5, false>(v, grid, smem, (int)B, u, k_lib, Tb, du, s);                \
           else launch_tiled_t<BSZ, false, 10, false>(v, grid, smem, (int)B, u, k_lib, Tb, du, s); }                 \
  } while (0)
  if (bs == 256) KIN_TILED_GO(256);
  else if (bs == 512) KIN_TILED_GO(512);
  else KIN_TILED_GO(1024);
#undef KIN_TILED_GO
  KIN_HIP(hipGetLastError());
}

}  // namespace kin
