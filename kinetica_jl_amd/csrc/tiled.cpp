// Host side of the tiled batched sweep: builds the library order described in tiled.hpp.
#include "tiled.hpp"

#include <algorithm>
#include <array>
#include <map>
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
  L.rec.clear(); L.kf.clear(); L.kr.clear();
  L.rowtab.clear(); L.seg_q.assign(1, 0);
  L.slot_of_reaction.assign(R, -1);
  for (int t = 0; t < T; t++) {
    std::stable_sort(seg[t].begin(), seg[t].end(), [&](int32_t a, int32_t b) { return cls(a) < cls(b); });
    const int32_t base = (int32_t)L.kf.size(), n = (int32_t)seg[t].size();
    std::vector<uint64_t> words(n);
    for (int32_t i = 0; i < n; i++) {
      const Rec& q = recs[seg[t][i]];
      uint64_t lab[4];
      for (int j = 0; j < 4; j++) {
        const int32_t sp = q.f[j];
        if (sp < 0) { lab[j] = (uint64_t)(h + (i & (TILED_DUMMY - 1))); continue; }
        const int32_t li = L.lib_of_species[sp];
        if (li < h) {
          const int c = i & 7;
          lab[j] = (copy_rank[sp] >= 0 && c > 0) ? (uint64_t)(h + TILED_DUMMY + copy_rank[sp] * TILED_COPIES + (c - 1)) : (uint64_t)li;
        } else {
          lab[j] = (uint64_t)(L.wbase + (li - L.win_off[t]));
        }
      }
      words[i] = lab[0] | (lab[1] << 14) | (lab[2] << 28) | (lab[3] << 42);
      L.kf.push_back(q.kf); L.kr.push_back(q.kr);
      L.slot_of_reaction[q.kf] = 2 * (base + i);
      if (q.kr >= 0) L.slot_of_reaction[q.kr] = 2 * (base + i) + 1;
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
    L.seginfo.push_back(base); L.seginfo.push_back(n); L.seginfo.push_back(rows_pad); L.seginfo.push_back(0);
    for (int32_t i = 0; i < rows_pad; i++) {
      L.rowtab.push_back(i < rows ? base + i * bs : -1);
      L.rowtab.push_back(i < rows ? std::min(bs, n - i * bs) : 0);
    }
    L.seg_q.push_back((int32_t)(L.rowtab.size() / 2));
  }
  L.ok = true;
  return L;
}

}  // namespace kin
