#include "network.hpp"

#include <algorithm>
#include <map>
#include <numeric>
#include <string>

namespace kin {

SegPlanHost build_seg_plan(int64_t n_rows, const int32_t* ptr, const int32_t* dst, const int32_t* a,
                           const int32_t* b, const float* c, bool skip_empty, const int32_t* aux,
                           std::vector<int32_t>* slot_of_entry) {
  SegPlanHost P;
  const bool prod = b != nullptr; // product plan: no coefficients, padding marked by b < 0
  if (slot_of_entry) slot_of_entry->assign((size_t)ptr[n_rows], -1);
  std::vector<int32_t> shorts, longs;
  for (int64_t i = 0; i < n_rows; i++) {
    int32_t len = ptr[i + 1] - ptr[i];
    if (len == 0 && skip_empty) continue;
    (len <= SegPlanHost::SHORT_MAX ? shorts : longs).push_back((int32_t)i);
  }
  // ELL groups: short rows ordered by length (stable; a counting sort - the lengths are 0 .. SHORT_MAX and the symbolic
  // products of the fused solve bring 0.5 M rows at 10k species, 4 M at 50k) so each 64-row group is padded to a
  // near-uniform width.
  {
    std::vector<size_t> start(SegPlanHost::SHORT_MAX + 2, 0);
    for (int32_t r : shorts) start[ptr[r + 1] - ptr[r] + 1]++;
    for (size_t l = 1; l < start.size(); l++) start[l] += start[l - 1];
    std::vector<int32_t> ordered(shorts.size());
    for (int32_t r : shorts) ordered[start[ptr[r + 1] - ptr[r]]++] = r;
    shorts.swap(ordered);
  }
  const size_t n_groups = (shorts.size() + 63) / 64;
  P.grp_off.reserve(n_groups + 1);
  P.grp_off.push_back(0);
  for (size_t g0 = 0; g0 < shorts.size(); g0 += 64) {      // sorted by length: the last row of a group is its widest
    const size_t g1 = std::min(shorts.size(), g0 + 64);
    P.grp_off.push_back(P.grp_off.back() + (ptr[shorts[g1 - 1] + 1] - ptr[shorts[g1 - 1]]));
  }
  {
    const size_t ell = (size_t)P.grp_off.back() * 64;
    if (a) P.ell_a.assign(ell, 0);
    if (b) P.ell_b.assign(ell, prod ? -1 : 0);
    if (!prod) P.ell_c.assign(ell, 0.0f);
    P.grp_dst.reserve(n_groups * 64); P.grp_aux.reserve(n_groups * 64);
  }
  for (size_t g = 0; g < n_groups; g++) {
    const size_t g0 = g * 64, g1 = std::min(shorts.size(), g0 + 64);
    const size_t ell_base = (size_t)P.grp_off[g] * 64;
    for (int lane = 0; lane < 64; lane++) {
      size_t q = g0 + lane;
      if (q >= g1) { P.grp_dst.push_back(-1); P.grp_aux.push_back(0); continue; }
      int32_t row = shorts[q];
      P.grp_dst.push_back(dst ? dst[row] : row);
      P.grp_aux.push_back(aux ? aux[row] : 0);
      for (int32_t e = ptr[row], col = 0; e < ptr[row + 1]; e++, col++) {
        size_t pos = ell_base + (size_t)col * 64 + lane;
        if (a) P.ell_a[pos] = a[e];
        if (slot_of_entry) (*slot_of_entry)[e] = (int32_t)pos;
        if (b) P.ell_b[pos] = b[e];
        if (!prod) P.ell_c[pos] = c ? c[e] : 1.0f;
      }
    }
  }
  // medium rows -> one segment (one wavefront); long rows -> one workgroup. Longest first: the long tasks start first.
  std::stable_sort(longs.begin(), longs.end(), [&](int32_t x, int32_t y) { return ptr[x + 1] - ptr[x] > ptr[y + 1] - ptr[y]; });
  const int32_t ell_total = P.grp_off.back() * 64;
  P.ell_total = ell_total;
  int32_t long_total = 0;
  for (int32_t row : longs) {
    const int32_t len = ptr[row + 1] - ptr[row];
    const int32_t out = dst ? dst[row] : row;
    const bool blk = len > SegPlanHost::SEG_LEN;
    (blk ? P.blk_beg : P.seg_beg).push_back(long_total);
    for (int32_t e = ptr[row]; e < ptr[row + 1]; e++) {
      if (a) P.long_a.push_back(a[e]);
      if (b) P.long_b.push_back(b[e]);
      if (!prod) P.long_c.push_back(c ? c[e] : 1.0f);
      if (slot_of_entry) (*slot_of_entry)[e] = ell_total + long_total;
      long_total++;
    }
    (blk ? P.blk_end : P.seg_end).push_back(long_total);
    (blk ? P.blk_dst : P.seg_dst).push_back(out);
    (blk ? P.blk_aux : P.seg_aux).push_back(aux ? aux[row] : 0);
  }
  P.long_total = long_total;
  return P;
}

static void fail(int code, const std::string& m) { throw KinError(code, m); }

NetworkHost compile_network(int64_t N, int64_t R, const int64_t* reac_ptr, const int64_t* reac_idx,
                            const int64_t* reac_sto, const int64_t* prod_ptr, const int64_t* prod_idx,
                            const int64_t* prod_sto, int index_base) {
  if (N <= 0 || R < 0) fail(ERR_INVALID_ARG, "n_species must be > 0 and n_reactions >= 0");
  if (N > (1ll << 30) || R > (1ll << 28)) fail(ERR_UNSUPPORTED, "network too large for int32 device indices");
  if (!reac_ptr || !prod_ptr || (R > 0 && (!reac_idx || !reac_sto || !prod_idx || !prod_sto)))
    fail(ERR_INVALID_ARG, "null topology array");
  if (index_base != 0 && index_base != 1) fail(ERR_INVALID_ARG, "index_base must be 0 or 1");
  NetworkHost H;
  H.N = N; H.R = R;
  H.x0.assign(R, -1); H.x1.assign(R, -1);
  H.y0.assign(R, -1); H.y1.assign(R, -1);
  H.slot_sp.assign(4 * R, -1); H.slot_co.assign(R, 0);

  struct Ent { int32_t rxn; float coef; };
  std::vector<std::vector<Ent>> by_species(N);
  // Jacobian contributions keyed by (row, col)
  struct JC { int32_t row, col, src; float coef; };
  std::vector<JC> jcs;
  jcs.reserve((size_t)R * 8);

  for (int64_t r = 0; r < R; r++) {
    int64_t a0 = reac_ptr[r], a1 = reac_ptr[r + 1], b0 = prod_ptr[r], b1 = prod_ptr[r + 1];
    if (a1 < a0 || b1 < b0) fail(ERR_INVALID_ARG, "ptr arrays must be non-decreasing");
    // rate operands, expanded by stoichiometry; the reference enforces molecularity <= 2
    // per side (src/exploration/network.jl:275-279)
    int32_t ops[2]; int nops = 0;
    std::map<int32_t, int> net;  // species -> net stoichiometric coefficient
    for (int64_t p = a0; p < a1; p++) {
      int64_t s = reac_idx[p] - index_base, st = reac_sto[p];
      if (s < 0 || s >= N) fail(ERR_INVALID_ARG, "reactant species index out of range in reaction " + std::to_string(r));
      if (st < 1) fail(ERR_INVALID_ARG, "reactant stoichiometry must be >= 1");
      for (int64_t q = 0; q < st; q++) {
        if (nops >= 2) fail(ERR_UNSUPPORTED, "reaction " + std::to_string(r) + ": more than 2 reactant molecules (max_molecularity = 2)");
        ops[nops++] = (int32_t)s;
      }
      net[(int32_t)s] -= (int)st;
    }
    if (nops == 0) fail(ERR_UNSUPPORTED, "reaction " + std::to_string(r) + " has no reactants");
    int nprod = 0;
    for (int64_t p = b0; p < b1; p++) {
      int64_t s = prod_idx[p] - index_base, st = prod_sto[p];
      if (s < 0 || s >= N) fail(ERR_INVALID_ARG, "product species index out of range in reaction " + std::to_string(r));
      if (st < 1 || st > 100) fail(ERR_INVALID_ARG, "product stoichiometry out of range");
      net[(int32_t)s] += (int)st;
      // product instances, expanded by stoichiometry (fixed-role records of the tiled sweep, tiled.cpp)
      for (int64_t q = 0; q < st; q++, nprod++) {
        if (nprod == 0) H.y0[r] = (int32_t)s;
        else if (nprod == 1) H.y1[r] = (int32_t)s;
      }
    }
    if (nprod > 2) H.products_le2 = false;
    H.x0[r] = ops[0];
    H.x1[r] = nops == 2 ? ops[1] : -1;
    // update slots
    int ns = 0; uint32_t packed = 0;
    for (auto& kv : net) {
      if (kv.second == 0) continue;  // e.g. an inert collider on both sides: multiplies the rate only
      if (ns >= 4) fail(ERR_UNSUPPORTED, "reaction " + std::to_string(r) + ": more than 4 distinct species");
      if (kv.second < -127 || kv.second > 127) fail(ERR_UNSUPPORTED, "net stoichiometry out of int8 range");
      H.slot_sp[4 * r + ns] = kv.first;
      packed |= (uint32_t)(uint8_t)(int8_t)kv.second << (8 * ns);
      by_species[kv.first].push_back({(int32_t)r, (float)kv.second});
      ns++;
    }
    H.slot_co[r] = (int32_t)packed;
    // Jacobian columns = distinct rate operands; drate index src = 2*r + w
    int ncols = (nops == 2 && ops[0] != ops[1]) ? 2 : 1;
    for (int w = 0; w < ncols; w++)
      for (auto& kv : net)
        if (kv.second != 0) jcs.push_back({kv.first, ops[w], (int32_t)(2 * r + w), (float)kv.second});
  }

  // ---- reversible-pair records (batched sweep). A reaction is "plain" when no species sits on
  // both sides; two plain reactions pair up when each one's net stoichiometry is the negative
  // of the other's and their operand lists are each other's product lists.
  if (N < 65535) {
    auto s16 = [](int32_t v) { return v < 0 ? 0xffffu : (uint32_t)v; };
    struct Key { int32_t sp[4]; int8_t co[4]; bool operator<(const Key& o) const {
      for (int j = 0; j < 4; j++) { if (sp[j] != o.sp[j]) return sp[j] < o.sp[j]; if (co[j] != o.co[j]) return co[j] < o.co[j]; }
      return false; } };
    auto key_of = [&](int64_t r, int sign, bool& plain) {
      Key kq; plain = true;
      int sumneg = 0;
      for (int j = 0; j < 4; j++) {
        kq.sp[j] = H.slot_sp[4 * r + j];
        int c = (int)(int8_t)((uint32_t)H.slot_co[r] >> (8 * j));
        kq.co[j] = (int8_t)(sign * c);
        if (kq.sp[j] >= 0 && c < 0) sumneg += -c;
      }
      // plain: the operands are exactly the negative-coefficient slots with matching multiplicity
      int nops = H.x1[r] >= 0 ? 2 : 1;
      if (sumneg != nops) plain = false;
      for (int j = 0; j < 4 && plain; j++) {
        int c = (int)(int8_t)((uint32_t)H.slot_co[r] >> (8 * j));
        if (kq.sp[j] >= 0 && c < 0) {
          int cnt = (H.x0[r] == kq.sp[j]) + (H.x1[r] == kq.sp[j]);
          if (cnt != -c) plain = false;
        }
      }
      // the reverse's operands (positive slots) must also number 1 or 2
      int sumpos = 0;
      for (int j = 0; j < 4; j++) { int c = (int)(int8_t)((uint32_t)H.slot_co[r] >> (8 * j)); if (c > 0) sumpos += c; }
      if (sumpos < 1 || sumpos > 2) plain = false;
      return kq;
    };
    std::map<Key, std::vector<int32_t>> waiting;   // forward key -> unmatched reactions
    std::vector<int32_t> partner(R, -1);
    std::vector<char> is_plain(R, 0);
    for (int64_t r = 0; r < R; r++) {
      bool plain;
      Key fwd = key_of(r, +1, plain);
      is_plain[r] = plain;
      if (!plain) continue;
      Key rev = key_of(r, -1, plain);
      auto it = waiting.find(rev);
      if (it != waiting.end() && !it->second.empty()) {
        int32_t q = it->second.back(); it->second.pop_back();
        partner[r] = q; partner[q] = (int32_t)r;
      } else waiting[fwd].push_back((int32_t)r);
    }
    for (int64_t r = 0; r < R; r++) {
      if (partner[r] >= 0 && partner[r] < r) continue;   // emitted with its forward
      H.pair_rec.push_back(s16(H.slot_sp[4 * r + 0]) | (s16(H.slot_sp[4 * r + 1]) << 16));
      H.pair_rec.push_back(s16(H.slot_sp[4 * r + 2]) | (s16(H.slot_sp[4 * r + 3]) << 16));
      H.pair_rec.push_back((uint32_t)H.slot_co[r]);
      // unpaired, non-plain records carry explicit operands; plain ones derive them from the signs
      H.pair_rec.push_back(is_plain[r] ? 0xffffffffu : (s16(H.x0[r]) | (s16(H.x1[r]) << 16)));
      H.pair_k.push_back((int32_t)r);
      H.pair_k.push_back(partner[r]);
    }
    H.pairs_adjacent = (R % 2 == 0) && (H.n_pairs() * 2 == R);
    for (int64_t p = 0; p < H.n_pairs() && H.pairs_adjacent; p++)
      if (H.pair_k[2 * p] != 2 * p || H.pair_k[2 * p + 1] != 2 * p + 1) H.pairs_adjacent = false;
    // 64-bit records of the register-resident sweep (kernels.hip: sweep_reg_kernel): four 14-bit species
    // labels with FIXED roles - fields 0, 1 = the forward reaction's reactant instances, fields 2, 3 = its
    // product instances (2A is listed as A, A) - so the kernel needs no coefficient or side decoding at
    // all: net = kf u0 u1 - kr u2 u3, du[0,1] -= net, du[2,3] += net. An unused field points at a per-lane
    // dummy entry N + (record index mod 64) whose u is 1.0 and whose du is never written out.
    H.pairs_block = !H.pairs_adjacent && (H.n_pairs() * 2 == R) && R > 0;
    for (int64_t p = 0; p < H.n_pairs() && H.pairs_block; p++)
      if (H.pair_k[2 * p] != p || H.pair_k[2 * p + 1] != H.n_pairs() + p) H.pairs_block = false;
    if ((H.pairs_adjacent || H.pairs_block) && N + 64 <= 16384) {
      bool ok = true;
      std::vector<int32_t> inst((size_t)4 * H.n_pairs());   // LDS entry per field before hub splitting
      for (int64_t p = 0; p < H.n_pairs() && ok; p++) {
        const uint32_t s01 = H.pair_rec[4 * p], s23 = H.pair_rec[4 * p + 1], co = H.pair_rec[4 * p + 2];
        const uint32_t sl[4] = {s01 & 0xffffu, s01 >> 16, s23 & 0xffffu, s23 >> 16};
        uint64_t side[2][2];
        int cnt[2] = {0, 0};
        for (int j = 0; j < 4 && ok; j++) {
          if (sl[j] == 0xffffu) continue;
          const int c = (int)(int8_t)(co >> (8 * j));
          if (c == 0 || c < -2 || c > 2) { ok = false; break; }
          const int sd = c < 0 ? 0 : 1;
          for (int q = 0; q < (c < 0 ? -c : c); q++) {
            if (cnt[sd] >= 2) { ok = false; break; }
            side[sd][cnt[sd]++] = sl[j];
          }
        }
        if (!ok) break;
        const uint64_t dummy = (uint64_t)N + (uint64_t)(p & 63);
        for (int sd = 0; sd < 2; sd++)
          while (cnt[sd] < 2) side[sd][cnt[sd]++] = dummy;
        for (int j = 0; j < 4; j++) inst[4 * p + j] = (int32_t)side[j / 2][j % 2];
      }
      if (ok) {
        // Split accumulators for the most referenced species: a same-address ds_add_f64 costs ~3 cycles per
        // lane (tools/lds_bank_probe.hip: 192 cycles for 64 lanes on one address against 13 conflict-free), and
        // under the Zipf wiring ~8 lanes of every 64-lane instruction hit the top species. Each of the top K
        // species gets 7 extra LDS entries behind the dummies; an occurrence in record p uses copy p mod 8
        // (copy 0 = the species' own entry). The kernel fills the copies' u and folds their du back per state.
        const int64_t P = H.n_pairs();
        const int64_t spare = (160 * 1024) / 16 - (N + 64);
        const int K = (int)std::max<int64_t>(0, std::min<int64_t>(32, spare / 7));
        std::vector<int64_t> refs(N, 0);
        for (int64_t q = 0; q < 4 * P; q++) if (inst[q] < N) refs[inst[q]]++;
        std::vector<int32_t> order(N);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return refs[x] > refs[y]; });
        std::vector<int32_t> rank_of(N, -1);
        H.sweep_copy_species.clear();
        for (int r = 0; r < K && r < N && refs[order[r]] >= 64; r++) {
          rank_of[order[r]] = r;
          for (int c = 1; c < 8; c++) H.sweep_copy_species.push_back(order[r]);
        }
        for (int64_t p = 0; p < P; p++) {
          uint64_t f[4];
          for (int j = 0; j < 4; j++) {
            const int32_t sp = inst[4 * p + j];
            const int c = (int)(p & 7);
            f[j] = (sp < N && rank_of[sp] >= 0 && c > 0) ? (uint64_t)(N + 64 + rank_of[sp] * 7 + (c - 1)) : (uint64_t)sp;
          }
          const uint64_t w = f[0] | (f[1] << 14) | (f[2] << 28) | (f[3] << 42);
          H.pair_rec64.push_back((uint32_t)w);
          H.pair_rec64.push_back((uint32_t)(w >> 32));
        }
      }
    }
    // ---- general fixed-role records (state fits LDS, pairing irregular): kernels.hip sweep_gen_kernel
    if (!H.pairs_adjacent && !H.pairs_block && (size_t)(N + 64) * 16 <= 160 * 1024) {
      const int64_t P = H.n_pairs();
      for (int64_t p = 0; p < P; p++) {
        const uint32_t s01 = H.pair_rec[4 * p], s23 = H.pair_rec[4 * p + 1], co = H.pair_rec[4 * p + 2], ops = H.pair_rec[4 * p + 3];
        const uint32_t sl[4] = {s01 & 0xffffu, s01 >> 16, s23 & 0xffffu, s23 >> 16};
        const uint32_t dummy = (uint32_t)N + (uint32_t)(p & 63);
        uint32_t side[2][2] = {{dummy, dummy}, {dummy, dummy}};
        int cnt[2] = {0, 0};
        if (ops == 0xffffffffu) {
          for (int j = 0; j < 4; j++) {
            if (sl[j] == 0xffffu) continue;
            const int c = (int)(int8_t)(co >> (8 * j));
            const int sd = c < 0 ? 0 : 1;
            for (int q = 0; q < (c < 0 ? -c : c) && cnt[sd] < 2; q++) side[sd][cnt[sd]++] = sl[j];
          }
        } else {
          H.gen_expl.push_back((int32_t)p);
        }
        H.gen_rec8.push_back(side[0][0] | (side[0][1] << 16));
        H.gen_rec8.push_back(side[1][0] | (side[1][1] << 16));
      }
    }
    // ---- large-N sweep tables (only when the state cannot live in LDS)
    if ((size_t)N * 16 > 160 * 1024 && N + 64 < 65535) {
      const int64_t P = H.n_pairs();
      const int32_t Hh = 10000;                       // hub species resident in LDS (2 x 80 kB)
      std::vector<int64_t> cnt(N, 0);
      for (int64_t p = 0; p < P; p++) {
        const uint32_t s01 = H.pair_rec[4 * p], s23 = H.pair_rec[4 * p + 1];
        const uint32_t sl[4] = {s01 & 0xffffu, s01 >> 16, s23 & 0xffffu, s23 >> 16};
        for (int j = 0; j < 4; j++) if (sl[j] != 0xffffu) cnt[sl[j]]++;
        const uint32_t ops = H.pair_rec[4 * p + 3];
        if (ops != 0xffffffffu) { cnt[ops & 0xffffu]++; if ((ops >> 16) != 0xffffu) cnt[ops >> 16]++; }
      }
      std::vector<int32_t> order(N);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cnt[a] > cnt[b]; });
      std::vector<char> is_hub(N, 0);
      for (int32_t q = 0; q < Hh; q++) is_hub[order[q]] = 1;
      std::vector<int32_t> label(N);
      H.big_spec_of_label.clear();
      for (int32_t sp = 0; sp < N; sp++) if (is_hub[sp]) { label[sp] = (int32_t)H.big_spec_of_label.size(); H.big_spec_of_label.push_back(sp); }
      for (int32_t sp = 0; sp < N; sp++) if (!is_hub[sp]) { label[sp] = (int32_t)H.big_spec_of_label.size(); H.big_spec_of_label.push_back(sp); }
      H.big_H = Hh;
      // tail operands by species id when the 16-bit label space has room for it (N + H + 64 <= 65535: C5 does)
      const bool by_species = (int64_t)N + Hh + 64 <= 65535;
      H.big_tail_by_species = by_species;
      auto relabel = [&](uint32_t v) { return v == 0xffffu ? 0xffffu : (uint32_t)label[v]; };
      H.big_rec.resize((size_t)4 * P);
      const int32_t TT = 2 * Hh;                      // tail labels accumulated per pass (the whole LDS)
      std::vector<std::vector<std::pair<uint32_t, uint32_t>>> tail(ceil_div(N - Hh, TT));
      for (int64_t p = 0; p < P; p++) {
        const uint32_t s01 = H.pair_rec[4 * p], s23 = H.pair_rec[4 * p + 1], ops = H.pair_rec[4 * p + 3];
        const uint32_t sl[4] = {relabel(s01 & 0xffffu), relabel(s01 >> 16), relabel(s23 & 0xffffu), relabel(s23 >> 16)};
        H.big_rec[4 * p + 0] = sl[0] | (sl[1] << 16);
        H.big_rec[4 * p + 1] = sl[2] | (sl[3] << 16);
        H.big_rec[4 * p + 2] = H.pair_rec[4 * p + 2];
        H.big_rec[4 * p + 3] = ops == 0xffffffffu ? ops : (relabel(ops & 0xffffu) | (relabel(ops >> 16) << 16));
        // 8-byte stream record: four 16-bit labels with fixed roles (fields 0, 1 reactant instances of the
        // forward reaction, 2, 3 its product instances); labels [Hh, Hh + 64) are the per-lane dummy entries,
        // tail labels are shifted up by 64. Records with explicit operands are all-dummy here and listed in big_expl.
        {
          const uint32_t dummy = (uint32_t)Hh + (uint32_t)(p & 63);
          uint32_t side[2][2] = {{dummy, dummy}, {dummy, dummy}};
          int cnt[2] = {0, 0};
          if (ops == 0xffffffffu) {
            for (int j = 0; j < 4; j++) {
              if (sl[j] == 0xffffu) continue;
              const int c = (int)(int8_t)(H.pair_rec[4 * p + 2] >> (8 * j));
              const int sd = c < 0 ? 0 : 1;
              for (int q = 0; q < (c < 0 ? -c : c) && cnt[sd] < 2; q++)
                side[sd][cnt[sd]++] = (int32_t)sl[j] < Hh ? sl[j] : (by_species ? (uint32_t)(Hh + 64 + H.big_spec_of_label[sl[j]]) : sl[j] + 64u);
            }
          } else {
            H.big_expl.push_back((int32_t)p);
          }
          H.big_rec8.push_back(side[0][0] | (side[0][1] << 16));
          H.big_rec8.push_back(side[1][0] | (side[1][1] << 16));
        }
        for (int j = 0; j < 4; j++) {
          if (sl[j] == 0xffffu || (int32_t)sl[j] < Hh) continue;
          const int32_t off = (int32_t)sl[j] - Hh;
          const uint32_t cf = (H.pair_rec[4 * p + 2] >> (8 * j)) & 0xffu;     // signed byte coefficient
          tail[off / TT].push_back({(uint32_t)p, (uint32_t)(off % TT) | (cf << 24)});
        }
      }
      H.big_tail_ptr.assign(1, 0);
      for (auto& tl : tail) {
        for (auto& e : tl) { H.big_tail_ent.push_back(e.first); H.big_tail_ent.push_back(e.second); }
        H.big_tail_ptr.push_back((int32_t)(H.big_tail_ent.size() / 2));
      }
    }
  }

  // species-major CSR
  H.sp_ptr.assign(N + 1, 0);
  for (int64_t i = 0; i < N; i++) H.sp_ptr[i + 1] = H.sp_ptr[i] + (int32_t)by_species[i].size();
  H.sp_rxn.reserve(H.sp_ptr[N]); H.sp_coef.reserve(H.sp_ptr[N]);
  for (int64_t i = 0; i < N; i++)
    for (auto& e : by_species[i]) { H.sp_rxn.push_back(e.rxn); H.sp_coef.push_back(e.coef); }

  // Jacobian pattern: sort contributions by (row, col, src); add the diagonal
  for (int64_t i = 0; i < N; i++) jcs.push_back({(int32_t)i, (int32_t)i, -1, 0.0f});
  std::sort(jcs.begin(), jcs.end(), [](const JC& p, const JC& q) {
    if (p.row != q.row) return p.row < q.row;
    if (p.col != q.col) return p.col < q.col;
    return p.src < q.src;
  });
  H.j_ptr.assign(N + 1, 0);
  H.j_diag.assign(N, -1);
  H.jc_ptr.push_back(0);
  for (size_t q = 0; q < jcs.size();) {
    size_t q2 = q;
    while (q2 < jcs.size() && jcs[q2].row == jcs[q].row && jcs[q2].col == jcs[q].col) q2++;
    int32_t e = (int32_t)H.j_col.size();
    H.j_col.push_back(jcs[q].col);
    H.j_ptr[jcs[q].row + 1]++;
    if (jcs[q].row == jcs[q].col) H.j_diag[jcs[q].row] = e;
    for (size_t t = q; t < q2; t++)
      if (jcs[t].src >= 0) { H.jc_src.push_back(jcs[t].src); H.jc_coef.push_back(jcs[t].coef); }
    H.jc_ptr.push_back((int32_t)H.jc_src.size());
    q = q2;
  }
  for (int64_t i = 0; i < N; i++) H.j_ptr[i + 1] += H.j_ptr[i];
  return H;
}

}  // namespace kin
