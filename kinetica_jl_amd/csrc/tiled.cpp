// Host side of the tiled batched sweep: builds the library order described in tiled.hpp.
#include "tiled.hpp"

#include <algorithm>
#include <array>
#include <map>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

namespace kin {

namespace {

struct UnionFind {
  std::vector<int32_t> par;
  explicit UnionFind(int32_t n) : par(n) { std::iota(par.begin(), par.end(), 0); }
  int32_t find(int32_t x) {
    while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; }
    return x;
  }
  void unite(int32_t a, int32_t b) {
    a = find(a); b = find(b);
    if (a != b) par[b] = a;
  }
};

struct Rec { int32_t f[4]; int32_t kf, kr; };   // species per field (-1: unused): reactant instances, product instances


// ---- placement of one segment's records (build_tiled below) ---------------------------------------------------------
struct SegCtx {
  const std::vector<Rec>& recs;
  const std::vector<int64_t>& cnt;              // references per species
  const std::vector<int32_t>& lib_of_species;
  const std::vector<int32_t>& copy_rank;        // split hubs: rank of the species among them, else -1
  int32_t h, wbase, win_off;                    // hubs, first window entry, first library species of this segment's window
  bool compact;                                 // records without a reverse get one rate-constant slot at the segment's end
  int scan;                                     // candidates in sight
  bool chunked;                                 // ... taken chunk by chunk instead of topped up after every placement
};
struct Placement {
  std::vector<int32_t> order;                   // record ids in library order
  std::vector<std::array<uint64_t, 4>> labs;    // their four LDS labels
  std::vector<char> flipped;                    // the record's reverse reaction takes the forward role
  int64_t slots = 0, conflicts = 0, deferred = 0;
};
inline int cls_of(const Rec& q) { return (q.f[1] >= 0 ? 1 : 0) | (q.f[3] >= 0 ? 2 : 0); }
inline int kind_of(const SegCtx& cx, const Rec& q) { return cx.compact && q.kr < 0 ? 4 : 0; }
// label options of field j of record p: a window entry, a hub entry, or a split hub's eight entries
int label_options(const SegCtx& cx, int32_t p, int j, uint64_t* opt) {
  const int32_t sp = cx.recs[p].f[j];
  if (sp < 0) return 0;
  const int32_t li = cx.lib_of_species[sp];
  if (li >= cx.h) { opt[0] = (uint64_t)(cx.wbase + (li - cx.win_off)); return 1; }
  opt[0] = (uint64_t)li;
  if (cx.copy_rank[sp] < 0) return 1;
  for (int c = 0; c < TILED_COPIES; c++) opt[1 + c] = (uint64_t)(cx.h + TILED_DUMMY + cx.copy_rank[sp] * TILED_COPIES + c);
  return 1 + TILED_COPIES;
}

// the plain order: as sorted (kind, class, then the caller's reaction order), a split hub's entries by position
Placement place_plain(const SegCtx& cx, const std::vector<int32_t>& sorted) {
  const int32_t n = (int32_t)sorted.size(), h = cx.h;
  Placement pl;
  pl.order = sorted; pl.labs.resize(n); pl.flipped.assign(n, 0);
  for (int32_t i = 0; i < n; i++)
    for (int j = 0; j < 4; j++) {
      uint64_t opt[1 + TILED_COPIES];
      const int no = label_options(cx, sorted[i], j, opt);
      pl.labs[i][j] = no == 0 ? (uint64_t)(h + (i & (TILED_DUMMY - 1))) : opt[no > 1 ? (i & 7) : 0];
    }
  return pl;
}

// the bank-aware order (see build_tiled): `sorted` = the segment's records by (kind, class, caller's reaction order)
Placement place_bank_aware(const SegCtx& cx, const std::vector<int32_t>& sorted) {
  const std::vector<Rec>& recs = cx.recs;
  const std::vector<int64_t>& cnt = cx.cnt;
  const int32_t n = (int32_t)sorted.size(), h = cx.h;
  const int SCAN = cx.scan;
  const bool chunked = cx.chunked;
  Placement pl;
  pl.order.resize(n); pl.labs.resize(n); pl.flipped.assign(n, 0);
  std::vector<int32_t>& placed = pl.order;
  std::vector<std::array<uint64_t, 4>>& labs = pl.labs;
  std::vector<char>& flipped = pl.flipped;
  int64_t& sched_slots = pl.slots; int64_t& sched_conflicts = pl.conflicts; int64_t& dbg_defer = pl.deferred;
  auto cls = [&](int32_t p) { return cls_of(recs[p]); };
  auto kind = [&](int32_t p) { return kind_of(cx, recs[p]); };
  auto options = [&](int32_t p, int j, uint64_t* opt) { return label_options(cx, p, j, opt); };
  // a candidate: per field of the record the banks its label options cover, mod 16 and mod 32 (0: field unused)
  struct Cand { int32_t p; uint16_t m16[4]; uint32_t m32[4]; int64_t hard; bool sides; int cls; };
  auto make_cand = [&](int32_t p) {
    Cand c{p, {0, 0, 0, 0}, {0, 0, 0, 0}, 0, false, cls(p)};
    for (int j = 0; j < 4; j++) {
      uint64_t opt[1 + TILED_COPIES];
      const int no = options(p, j, opt);
      for (int o = 0; o < no; o++) { c.m16[j] |= (uint16_t)(1u << (opt[o] & 15)); c.m32[j] |= 1u << (opt[o] & 31); }
      if (no > 0) c.hard = std::max(c.hard, cnt[recs[p].f[j]]);
    }
    // forward and reverse reaction of a pair may change roles where that keeps the record's class
    c.sides = recs[p].kr >= 0 && (recs[p].f[1] >= 0) == (recs[p].f[3] >= 0);
    return c;
  };
  // orientation o: bit 0 = the two reactant fields swapped, bit 1 = the two product fields swapped, bit 2 = the
  // sides swapped (the reverse reaction becomes the record's forward one); src[j] = the record field that goes to j
  auto sources = [](int o, int* src) {
    const int s0 = (o & 4) ? 2 : 0, s1 = (o & 4) ? 0 : 2;
    src[0] = s0 + (o & 1); src[1] = s0 + ((o & 1) ^ 1);
    src[2] = s1 + ((o >> 1) & 1); src[3] = s1 + (((o >> 1) & 1) ^ 1);
  };
  bool present[8] = {false, false, false, false, false, false, false, false};
  for (int32_t x = 0; x < n; x++) present[kind(sorted[x]) | cls(sorted[x])] = true;
  // a record may also ride in the wavefronts of a class that has all its fields (they execute the instructions of the
  // fields it lacks anyway, on dummy entries): what cannot be placed without a conflict at the end of a class's run -
  // when the candidates run out, the last lanes of a group rarely find their one free bank - waits for the next such run
  auto later_home = [&](int x, int c, int kd) {   // (among the runs of the same kind)
    for (int c2 = c + 1; c2 < 4; c2++) if (present[kd | c2] && (x & ~c2) == 0) return true;
    return false;
  };
  std::vector<Cand> deferred;
  uint32_t used[4] = {0, 0, 0, 0}, used32[4] = {0, 0, 0, 0};
  int32_t i = 0;
  for (int32_t a = 0; a < n;) {
    int32_t b = a;
    const int c_run = cls(sorted[a]), k_run = kind(sorted[a]);
    while (b < n && cls(sorted[b]) == c_run && kind(sorted[b]) == k_run) b++;
    const bool live[4] = {true, (c_run & 1) != 0, true, (c_run & 2) != 0};
    // the candidates in sight, taken in the caller's reaction order (the conversion of rate constants to the library order
    // then gathers from a narrow range per wavefront)
    std::vector<Cand> pool;
    pool.reserve(SCAN + deferred.size());
    for (size_t d = 0; d < deferred.size();)
      if ((deferred[d].cls & ~c_run) == 0 && kind(deferred[d].p) == k_run) { pool.push_back(deferred[d]); deferred[d] = deferred.back(); deferred.pop_back(); }
      else d++;
    int32_t next = a;
    size_t cur = 0;
    for (;;) {
      if (!chunked || pool.empty()) while (pool.size() < (size_t)SCAN && next < b) pool.push_back(make_cand(sorted[next++]));
      if (pool.empty()) break;
      if ((i & 15) == 0) used[0] = used[1] = used[2] = used[3] = 0;
      if ((i & 31) == 0) used32[0] = used32[1] = used32[2] = used32[3] = 0;
      // Among them, in any orientation, one that finds a free bank in every field - for the ds_add_f64 of its 16-lane
      // group first (weight 4), for the ds_read_b64 of its 32-lane half second; of the first few that do, the one with
      // the most referenced species (the hubs' records are the hard ones to place).
      uint32_t fr[4], fr32[4];
      for (int j = 0; j < 4; j++) { fr[j] = ~used[j] & 0xffffu; fr32[j] = ~used32[j]; }
      size_t best = cur % pool.size(); int best_cost = 1 << 20, best_or = 0, fits = 0; int64_t best_hard = -1;
      const int enough = (i & 15) >= 12 ? 1 : 6;   // (the last lanes of a group take the first record that fits: few do)
      for (size_t c = 0; c < pool.size() && fits < enough; c++) {
        const size_t q = (cur + c) % pool.size();
        const Cand& k = pool[q];
        // (quick look: can anything of the record sit in fields 0 and 2 at all? most candidates fail here when the
        // group is nearly full)
        if (best_cost <= 4) {
          const bool a0 = ((k.m16[0] | k.m16[1]) & fr[0]) != 0, a2 = ((k.m16[2] | k.m16[3]) & fr[2]) != 0;
          const bool b0 = k.sides && ((k.m16[2] | k.m16[3]) & fr[0]) != 0, b2 = k.sides && ((k.m16[0] | k.m16[1]) & fr[2]) != 0;
          if (!((a0 && a2) || (b0 && b2))) continue;
        }
        // (the cost separates: per side assignment, the better order of the reactant pair + the better order of the
        // product pair)
        auto field_cost = [&](int sj, int j) {
          if (!k.m16[sj]) return (live[j] && !fr[j]) ? 4 : 0;
          if (!(k.m16[sj] & fr[j])) return 4;
          return (k.m32[sj] & fr32[j]) ? 0 : 1;
        };
        int q_cost = 1 << 20;
        for (int sd = 0; sd < (k.sides ? 2 : 1); sd++) {
          const int s0 = sd ? 2 : 0, s1 = sd ? 0 : 2;
          int cr = field_cost(s0, 0) + field_cost(s0 + 1, 1), o_r = 0;
          if (k.m16[s0 + 1] && cr > 0) { const int c2 = field_cost(s0 + 1, 0) + field_cost(s0, 1); if (c2 < cr) { cr = c2; o_r = 1; } }
          int cp = field_cost(s1, 2) + field_cost(s1 + 1, 3), o_p = 0;
          if (k.m16[s1 + 1] && cp > 0) { const int c2 = field_cost(s1 + 1, 2) + field_cost(s1, 3); if (c2 < cp) { cp = c2; o_p = 2; } }
          const int cost = cr + cp;
          q_cost = std::min(q_cost, cost);
          if (cost < best_cost || (cost == best_cost && k.hard > best_hard)) { best_cost = cost; best = q; best_or = o_r | o_p | (sd ? 4 : 0); best_hard = k.hard; }
        }
        fits += q_cost == 0;
      }
      if (best_cost >= 4 && next == b) {
        // the end of the run: whoever has a later home goes there
        size_t kept = 0;
        for (size_t q = 0; q < pool.size(); q++)
          if (later_home(pool[q].cls, c_run, k_run)) deferred.push_back(pool[q]); else pool[kept++] = pool[q];
        if (kept < pool.size()) { dbg_defer += pool.size() - kept; pool.resize(kept); cur = 0; continue; }
      }
      const int32_t p = pool[best].p;
      int src[4];
      sources(best_or, src);
      for (int j = 0; j < 4; j++) {
        uint64_t opt[TILED_DUMMY];
        int no = options(p, src[j], opt);
        if (no == 0) {
          if (!live[j]) { labs[i][j] = (uint64_t)(h + (i & (TILED_DUMMY - 1))); continue; }
          // a field the record lacks but its wavefront executes: any dummy entry on a free bank
          for (int d = 0; d < TILED_DUMMY; d++) opt[d] = (uint64_t)(h + ((i + d) & (TILED_DUMMY - 1)));
          no = TILED_DUMMY;
        }
        int pick = -1;
        for (int o = 0; o < no && pick < 0; o++)
          if (!(used[j] >> (opt[o] & 15) & 1u) && !(used32[j] >> (opt[o] & 31) & 1u)) pick = o;
        for (int o = 0; o < no && pick < 0; o++) if (!(used[j] >> (opt[o] & 15) & 1u)) pick = o;
        sched_slots++;
        if (pick < 0) { pick = no > 1 ? i % no : 0; sched_conflicts++; }
        labs[i][j] = opt[pick];
        used[j] |= 1u << (opt[pick] & 15);
        used32[j] |= 1u << (opt[pick] & 31);
      }
      flipped[i] = (best_or & 4) != 0;
      placed[i++] = p;
      pool[best] = pool.back();
      pool.pop_back();
      cur = best + 1;
    }
    a = b;
  }
  return pl;
}

}  // namespace

TiledHost build_tiled(const NetworkHost& H, int bs, int h_force) {
  TiledHost L;
  const int32_t N = (int32_t)H.N, R = (int32_t)H.R;
  L.N = N; L.R = R; L.BS = bs;
  if (!H.products_le2) { L.why = "a reaction has more than two product molecules"; return L; }
  if (H.N >= (1 << 30)) { L.why = "too many species"; return L; }

  // ---- records: a reaction and its exact reverse (reactant multiset = the other's product multiset) share one
  auto key_of = [&](int32_t r, bool reversed) {
    std::array<int32_t, 4> k = {H.x0[r], H.x1[r], H.y0[r], H.y1[r]};
    // canonical order inside each side: (a, b) with a <= b, a single molecule as (a, -1)
    if (k[1] >= 0 && k[1] < k[0]) std::swap(k[0], k[1]);
    if (k[3] >= 0 && k[3] < k[2]) std::swap(k[2], k[3]);
    if (reversed) { std::swap(k[0], k[2]); std::swap(k[1], k[3]); }
    return k;
  };
  std::vector<Rec> recs;
  recs.reserve((size_t)R / 2 + 16);
  {
    std::map<std::array<int32_t, 4>, std::vector<int32_t>> waiting;   // forward key -> records still without a reverse
    for (int32_t r = 0; r < R; r++) {
      if (H.y0[r] >= 0) {   // (a reaction without products has no reverse)
        auto it = waiting.find(key_of(r, true));
        if (it != waiting.end() && !it->second.empty()) {
          recs[it->second.back()].kr = r;
          it->second.pop_back();
          continue;
        }
      }
      recs.push_back(Rec{{H.x0[r], H.x1[r], H.y0[r], H.y1[r]}, r, -1});
      if (H.y0[r] >= 0) waiting[key_of(r, false)].push_back((int32_t)recs.size() - 1);
    }
  }
  const int32_t P = (int32_t)recs.size();
  L.P = P;

  // ---- popularity
  std::vector<int64_t> cnt(N, 0);
  for (const Rec& q : recs)
    for (int j = 0; j < 4; j++) if (q.f[j] >= 0) cnt[q.f[j]]++;
  std::vector<int32_t> order(N);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cnt[a] > cnt[b]; });
  std::vector<int32_t> rank(N);
  for (int32_t i = 0; i < N; i++) rank[order[i]] = i;

  // ---- split accumulators for the most referenced species (a same-address ds_add_f64 serialises, ~3 cycles per lane)
  // (KIN_TILED_ENTRIES shrinks the on-chip capacity the layout is built for: tests exercise windows on small networks)
  int E_cap = TILED_LDS_ENTRIES;
  if (const char* e = getenv("KIN_TILED_ENTRIES")) E_cap = std::max(256, std::min(TILED_LDS_ENTRIES, atoi(e)));
  int K = 0;
  while (K < 25 && K < N && cnt[order[K]] >= 256) K++;
  // a state that just fits keeps as many split hubs as there is room for (N = 10 000: 16 of them)
  if (N + TILED_DUMMY + K * TILED_COPIES > E_cap && N + TILED_DUMMY <= E_cap) K = std::max(0, (E_cap - N - TILED_DUMMY) / TILED_COPIES);
  L.n_copy = K * TILED_COPIES;

  // ---- hubs and windows
  std::vector<int32_t> window_of(N, -1);     // tail species -> window
  std::vector<int32_t> seg_of_rec(P, -1);
  int32_t h = N, T = 1, w = 0;
  if (N + TILED_DUMMY + L.n_copy > E_cap) {
    // smallest hub set whose tail falls apart into components much smaller than a window (fewest windows)
    std::vector<int32_t> comp_root;
    bool found = false;
    std::vector<int32_t> cands;
    if (h_force > 0) cands.push_back(h_force);
    else for (int pc = 30; pc <= 85; pc += 5) cands.push_back((int32_t)((int64_t)E_cap * pc / 100));
    for (int32_t hc : cands) {
      // the windows start at an EVEN label (wbase below is rounded up) and the label space ends at an even E: the window
      // capacity is what is left behind the rounded base, rounded down - a completely full window still ends at E <= E_cap
      int32_t wb_c = hc + TILED_DUMMY + L.n_copy;
      wb_c += wb_c & 1;
      const int32_t wc = ((E_cap & ~1) - wb_c) & ~1;
      if (wc < 64) continue;
      UnionFind uf(N);
      for (const Rec& q : recs) {
        int32_t first = -1;
        for (int j = 0; j < 4; j++) {
          const int32_t sp = q.f[j];
          if (sp < 0 || rank[sp] < hc) continue;
          if (first < 0) first = sp; else uf.unite(first, sp);
        }
      }
      std::vector<int32_t> size(N, 0);
      int32_t biggest = 0;
      for (int32_t sp = 0; sp < N; sp++) if (rank[sp] >= hc) biggest = std::max(biggest, ++size[uf.find(sp)]);
      if (biggest * 3 <= wc) {
        h = hc; w = wc; found = true;
        comp_root.assign(N, -1);
        for (int32_t sp = 0; sp < N; sp++) if (rank[sp] >= hc) comp_root[sp] = uf.find(sp);
        break;
      }
    }
    if (!found) { L.why = "the rarely referenced species do not decompose into window-sized groups"; return L; }
    // components -> windows: largest first, into the window with room that has the fewest records so far
    std::map<int32_t, std::vector<int32_t>> members;
    for (int32_t sp = 0; sp < N; sp++) if (comp_root[sp] >= 0) members[comp_root[sp]].push_back(sp);
    std::map<int32_t, int64_t> comp_recs;
    for (const Rec& q : recs)
      for (int j = 0; j < 4; j++) if (q.f[j] >= 0 && comp_root[q.f[j]] >= 0) { comp_recs[comp_root[q.f[j]]]++; break; }
    std::vector<int32_t> roots;
    for (auto& kv : members) roots.push_back(kv.first);
    std::stable_sort(roots.begin(), roots.end(), [&](int32_t a, int32_t b) { return members[a].size() > members[b].size(); });
    T = (int32_t)ceil_div(N - h, w);
    std::vector<int32_t> fill(T, 0);
    std::vector<int64_t> load(T, 0);
    for (int32_t root : roots) {
      const int32_t sz = (int32_t)members[root].size();
      int best = -1;
      for (int t = 0; t < T; t++)
        if (fill[t] + sz <= w && (best < 0 || load[t] < load[best])) best = t;
      if (best < 0) { best = T++; fill.push_back(0); load.push_back(0); }
      fill[best] += sz;
      load[best] += comp_recs[root];
      for (int32_t sp : members[root]) window_of[sp] = best;
    }
    // records: the window of their tail species; records on hubs only fill the segments up evenly
    std::vector<int64_t> nrec(T, 0);
    std::vector<int32_t> free_recs;
    for (int32_t p = 0; p < P; p++) {
      int32_t wdw = -1;
      for (int j = 0; j < 4; j++) if (recs[p].f[j] >= 0 && window_of[recs[p].f[j]] >= 0) wdw = window_of[recs[p].f[j]];
      seg_of_rec[p] = wdw;
      if (wdw >= 0) nrec[wdw]++; else free_recs.push_back(p);
    }
    // Hub-only records first fill the padding of the segments' last batches (a segment's rows are padded to whole
    // batches of `quantum` rows, at least TILED_MIN_ROWS), then go in whole batches to the segment with the fewest
    // records: the iteration space the kernel walks is then the smallest the window records allow (C5: 124 rows of
    // 1 024 records for 122.07 rows of work; filling up to P / T per segment and padding to 8 rows made it 144).
    int32_t wmax_now = 0;
    for (int t = 0; t < T; t++) wmax_now = std::max(wmax_now, fill[t]);
    L.row_quantum = std::max(h, wmax_now) <= 5 * bs ? 4 : 2;
    const int64_t Q = (int64_t)bs * L.row_quantum;
    size_t fr = 0;
    for (int t = 0; t < T; t++) {
      const int64_t cap = std::max<int64_t>((int64_t)TILED_MIN_ROWS * bs, ceil_div(nrec[t], Q) * Q);
      while (fr < free_recs.size() && nrec[t] < cap) { seg_of_rec[free_recs[fr++]] = t; nrec[t]++; }
    }
    while (fr < free_recs.size()) {
      int best = 0;
      for (int t = 1; t < T; t++) if (nrec[t] < nrec[best]) best = t;
      for (int64_t i = 0; i < Q && fr < free_recs.size(); i++) { seg_of_rec[free_recs[fr++]] = best; nrec[best]++; }
    }
  } else {
    L.row_quantum = h <= 5 * bs ? 4 : 2;
    std::fill(seg_of_rec.begin(), seg_of_rec.end(), 0);
  }
  if (T > TILED_MAX_SEG) { L.why = "too many windows"; return L; }
  L.h = h; L.T = T;
  L.wbase = h + TILED_DUMMY + L.n_copy;
  L.wbase += L.wbase & 1;

  // ---- library species order: hubs in species order, then window after window
  L.species_of_lib.clear();
  L.lib_of_species.assign(N, -1);
  for (int32_t sp = 0; sp < N; sp++) if (h == N || rank[sp] < h) { L.lib_of_species[sp] = (int32_t)L.species_of_lib.size(); L.species_of_lib.push_back(sp); }
  L.win_off.assign(T, h); L.win_cnt.assign(T, 0);
  if (h < N) {
    std::vector<std::vector<int32_t>> ws(T);
    for (int32_t sp = 0; sp < N; sp++) if (window_of[sp] >= 0) ws[window_of[sp]].push_back(sp);
    for (int t = 0; t < T; t++) {
      L.win_off[t] = (int32_t)L.species_of_lib.size();
      L.win_cnt[t] = (int32_t)ws[t].size();
      for (int32_t sp : ws[t]) { L.lib_of_species[sp] = (int32_t)L.species_of_lib.size(); L.species_of_lib.push_back(sp); }
    }
  }
  L.identity = true;
  for (int32_t i = 0; i < N; i++) if (L.species_of_lib[i] != i) { L.identity = false; break; }
  if (!L.identity) {
    // staging tables of the layout conversion (tiled.hpp): groups = hubs, then the windows; each is ascending in the caller's index
    std::vector<int32_t> g_lo{0}, g_hi{h};
    for (int t = 0; t < T; t++) { g_lo.push_back(L.win_off[t]); g_hi.push_back(L.win_off[t] + L.win_cnt[t]); }
    L.stage_lib.clear(); L.stage_off.clear();
    L.stage_lib.reserve(N); L.stage_off.reserve(N);
    std::vector<int32_t> cur(g_lo);     // next library index of every group not yet emitted
    for (int32_t p0 = 0; p0 < N; p0 += TILED_PIECE) {
      const int32_t p1 = std::min<int32_t>(N, p0 + TILED_PIECE);
      for (size_t g = 0; g < cur.size(); g++)
        while (cur[g] < g_hi[g] && L.species_of_lib[cur[g]] < p1) {
          L.stage_lib.push_back(cur[g]);
          L.stage_off.push_back(L.species_of_lib[cur[g]] - p0);
          cur[g]++;
        }
      if ((int32_t)L.stage_lib.size() != p1) { L.why = "internal: staging tables of the layout conversion do not cover a piece"; return L; }
    }
  }
  int32_t wmax = 0;
  for (int t = 0; t < T; t++) wmax = std::max(wmax, L.win_cnt[t]);
  L.E = L.wbase + wmax;
  L.E += L.E & 1;
  if (L.E > (1 << 14)) { L.why = "LDS label space exceeded"; return L; }
  if (h < N && L.E > E_cap) { L.why = "internal: a window overflows the on-chip label space"; return L; }

  // ---- split hubs
  std::vector<int32_t> copy_rank(N, -1);
  L.copy_src.clear();
  for (int r = 0; r < K; r++) {
    copy_rank[order[r]] = r;
    for (int c = 0; c < TILED_COPIES; c++) L.copy_src.push_back(L.lib_of_species[order[r]]);
  }

  // ---- records in library order: segment after segment, inside a segment sorted by which optional fields they use
  // (second reactant, second product) so that whole wavefronts can skip the LDS operations of an unused field
  std::vector<std::vector<int32_t>> seg(T);
  for (int32_t p = 0; p < P; p++) seg[seg_of_rec[p]].push_back(p);
  auto cls = [&](int32_t p) { return (recs[p].f[1] >= 0 ? 1 : 0) | (recs[p].f[3] >= 0 ? 2 : 0); };
  L.rec.clear(); L.kf.clear(); L.kr.clear(); L.kslot.clear(); L.seg_k.clear();
  L.KL = 0; L.has_singles = false; L.pad_slots.clear();
  L.rowtab.clear(); L.seg_q.assign(1, 0);
  L.slot_of_reaction.assign(R, -1);
  // Inside a class the order is free, and so are the order of a record's two reactants / two products, which reaction of a
  // pair plays the forward role, and the accumulator entry a split hub's occurrence uses: all are chosen so that the 16
  // records of a lane group (segment positions 16 g .. 16 g + 15) hit 16 different banks in each field. A ds_add_f64 is
  // executed in four groups of 16 contiguous lanes over 16 banks of 8 bytes (bank = label mod 16), a ds_read_b64 in two
  // halves of 32 lanes over 32 banks (tools/lds_group_probe.hip: ds_add_f64 3.4 ns per wave instruction when a group's
  // labels differ mod 16, 9-11 ns for random labels, 8.4 ns for two lanes of a group on one address; lanes of different
  // groups never conflict). Unscheduled, the labels of the synthetic CRNs cost 3.0 LDS cycles per group and field; placed
  // from a reservoir of 256-2 048 candidates 1.2-1.5, placed inside chunks of 64 records (the default) 2.0. The sweeps are as
  // fast with the one as with the other (profiles/r04_tiled_schedule_ab.txt: after the first third of the conflicts is gone
  // the kernel no longer waits for the LDS), and the chunked order keeps kin_rates_to_lib_dev's gathers as local as the plain
  // order does (C3: 0.79 ms plain, 0.82 chunked, 1.13 with the reservoir of 256).
  // KIN_TILED_SCHEDULE=0 keeps the plain order (A/B), KIN_TILED_SCAN sets the number of candidates in sight.
  const bool schedule = !(getenv("KIN_TILED_SCHEDULE") && atoi(getenv("KIN_TILED_SCHEDULE")) == 0);
  // Records without a reverse reaction (what the low-k cutoff leaves of a pair, solve_utils.jl:213-245) come last in their
  // segment and take ONE rate-constant slot there; the records before them - all pairs, filled up to whole wavefronts with
  // single ones - take two (tiled.hpp: seg_k). Only when that shortens the k row noticeably: a network that has all its
  // pairs keeps the plain layout and the plain instantiation of the kernel.
  int64_t n_single = 0;
  for (const Rec& q : recs) n_single += q.kr < 0;
  const bool compact = n_single * 20 >= (int64_t)P && !(getenv("KIN_TILED_SINGLES") && atoi(getenv("KIN_TILED_SINGLES")) == 0);
  auto kind = [&](int32_t p) { return compact && recs[p].kr < 0 ? 4 : 0; };
  int SCAN = 64;           // candidates in sight per position
  if (const char* e = getenv("KIN_TILED_SCAN")) SCAN = std::max(16, atoi(e));
  // chunked (default): the candidates are the next SCAN records in the caller's reaction order and all of them are placed before
  // the next SCAN come in sight - the library order is the caller's, permuted inside chunks of one wavefront's records;
  // KIN_TILED_CHUNKED=0: a reservoir that is topped up after every placement (fewer conflicts, a wider scramble)
  const bool chunked = !(getenv("KIN_TILED_CHUNKED") && atoi(getenv("KIN_TILED_CHUNKED")) == 0);
  int64_t sched_slots = 0, sched_conflicts = 0, dbg_defer = 0;
  double dbg_sec = 0.0;
  for (int t = 0; t < T; t++) {
    std::stable_sort(seg[t].begin(), seg[t].end(), [&](int32_t a, int32_t b) { return (kind(a) | cls(a)) < (kind(b) | cls(b)); });
    const int32_t base = (int32_t)L.kf.size(), n = (int32_t)seg[t].size();
    const SegCtx cx{recs, cnt, L.lib_of_species, copy_rank, h, L.wbase, L.win_off[t], compact, SCAN, chunked};
    const auto t_s0 = std::chrono::steady_clock::now();
    Placement pl = schedule ? place_bank_aware(cx, seg[t]) : place_plain(cx, seg[t]);
    sched_slots += pl.slots; sched_conflicts += pl.conflicts; dbg_defer += pl.deferred;
    const std::vector<int32_t>& placed = pl.order;
    const std::vector<std::array<uint64_t, 4>>& labs = pl.labs;
    const std::vector<char>& flipped = pl.flipped;
    dbg_sec += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_s0).count();
    seg[t] = placed;
    // rate-constant slots: the first n2 records two each (n2 = the pairs, rounded up to whole wavefronts), the rest one
    int32_t n_pairs = 0;
    for (int32_t i = 0; i < n; i++) n_pairs += kind(seg[t][i]) == 0;
    const int32_t n2 = compact ? std::min<int32_t>(n, (int32_t)ceil_div(n_pairs, 64) * 64) : n;
    const int32_t koff = (int32_t)L.KL;
    L.seg_k.push_back(koff); L.seg_k.push_back(n2);
    L.KL += 2 * (int64_t)n2 + (n - n2);
    if (L.KL & 1) L.pad_slots.push_back((int32_t)L.KL++);
    L.has_singles = L.has_singles || n2 < n;
    std::vector<uint64_t> words(n);
    for (int32_t i = 0; i < n; i++) {
      const Rec& q = recs[seg[t][i]];
      const uint64_t* lab = labs[i].data();
      words[i] = lab[0] | (lab[1] << 14) | (lab[2] << 28) | (lab[3] << 42);
      const int32_t rf = flipped[i] ? q.kr : q.kf, rr = flipped[i] ? q.kf : q.kr;
      L.kf.push_back(rf); L.kr.push_back(rr);
      const int32_t slot = koff + (i < n2 ? 2 * i : n2 + i);
      L.slot_of_reaction[rf] = slot;
      if (rr >= 0) {
        if (i >= n2) { L.why = "internal: a pair among the single-slot records"; return L; }
        L.slot_of_reaction[rr] = slot + 1;
      }
      L.kslot.push_back(i < n2 ? slot : ~slot);
    }
    // flags of a 64-record group (one wavefront's records of a row) = OR over its records: a group of one kind skips
    // the unused fields, a mixed group sends them to the dummy entries
    for (int32_t g0 = 0; g0 < n; g0 += 64) {
      uint64_t fl = 0;
      for (int32_t i = g0; i < std::min(n, g0 + 64); i++) fl |= (uint64_t)cls(seg[t][i]);
      for (int32_t i = g0; i < std::min(n, g0 + 64); i++) words[i] |= fl << 56;
    }
    for (int32_t i = 0; i < n; i++) { L.rec.push_back((uint32_t)words[i]); L.rec.push_back((uint32_t)(words[i] >> 32)); }
    const int32_t rows = (int32_t)ceil_div(n, bs);
    // (at least one group of iteration rows, also for a window without records: the kernel's producer relies on it)
    const int32_t rows_pad = std::max<int32_t>(TILED_MIN_ROWS, (int32_t)ceil_div(rows, L.row_quantum) * L.row_quantum);
    L.seginfo.push_back(base); L.seginfo.push_back(n); L.seginfo.push_back(rows_pad); L.seginfo.push_back(n2);
    for (int32_t i = 0; i < rows_pad; i++) {
      L.rowtab.push_back(i < rows ? base + i * bs : -1);
      L.rowtab.push_back(i < rows ? std::min(bs, n - i * bs) : 0);
    }
    L.seg_q.push_back((int32_t)(L.rowtab.size() / 2));
  }
  L.sched_slots = sched_slots; L.sched_conflicts = sched_conflicts;
  if (getenv("KIN_TILED_DEBUG")) fprintf(stderr, "[tiled] %d records, %d segments: %lld of %lld label slots share a bank with an earlier lane of their group\n",
                                         P, T, (long long)sched_conflicts, (long long)sched_slots),
    fprintf(stderr, "[tiled] %lld records moved to the wavefronts of a later class, %.3f s in the record order\n", (long long)dbg_defer, dbg_sec);
  L.ok = true;
  return L;
}

}  // namespace kin
