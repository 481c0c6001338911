#include "lu.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <map>
#include <numeric>
#include <thread>

namespace kin {

namespace {

// position of `target` in the sorted neighbour list of pivot p
inline int32_t find_entry(const std::vector<int32_t>& ent_ptr, const std::vector<int32_t>& nbr, int32_t p, int32_t target) {
  auto b = nbr.begin() + ent_ptr[p], e = nbr.begin() + ent_ptr[p + 1];
  auto it = std::lower_bound(b, e, target);
  if (it == e || *it != target) throw KinError(ERR_DEVICE, "internal: missing fill entry in symbolic LU");
  return (int32_t)(it - nbr.begin());
}

struct Triple { int64_t t; int32_t a, b; };

SegPlanHost plan_from_triples(std::vector<Triple>& tr, bool with_aux_diag, int64_t aux_base, int64_t dst_base) {
  (void)with_aux_diag; (void)aux_base; (void)dst_base;
  std::sort(tr.begin(), tr.end(), [](const Triple& x, const Triple& y) {
    if (x.t != y.t) return x.t < y.t;
    if (x.a != y.a) return x.a < y.a;
    return x.b < y.b;
  });
  std::vector<int32_t> ptr{0}, dst, a, b;
  a.reserve(tr.size()); b.reserve(tr.size());
  for (size_t q = 0; q < tr.size();) {
    size_t q2 = q;
    while (q2 < tr.size() && tr[q2].t == tr[q].t) { a.push_back(tr[q2].a); b.push_back(tr[q2].b); q2++; }
    dst.push_back((int32_t)tr[q].t);
    ptr.push_back((int32_t)a.size());
    q = q2;
  }
  return build_seg_plan((int64_t)dst.size(), ptr.data(), dst.data(), a.data(), b.data(), nullptr, true);
}

}  // namespace

void SparseLU::analyze(int32_t n_, const std::vector<int32_t>& j_ptr, const std::vector<int32_t>& j_col,
                       const LUOptions& opt, hipStream_t s) {
  n = n_;
  nnzJ = (int64_t)j_col.size();
  const bool dbg_time = getenv("KIN_LU_DEBUG") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_time) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[lu] analyze: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  // ---- symmetric adjacency of J + J^T without self loops
  std::vector<std::vector<int32_t>> adj(n);
  for (int32_t i = 0; i < n; i++)
    for (int32_t e = j_ptr[i]; e < j_ptr[i + 1]; e++) {
      int32_t j = j_col[e];
      if (j != i) { adj[i].push_back(j); adj[j].push_back(i); }
    }
  std::vector<char> hub(n, 0);
  for (int32_t i = 0; i < n; i++) {
    auto& a = adj[i];
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
    if ((int)a.size() >= opt.hub_degree) hub[i] = 1;
  }
  for (int32_t i = 0; i < n; i++)
    if (hub[i]) std::vector<int32_t>().swap(adj[i]);  // hub lists are never needed

  // ---- rounds of independent-set elimination
  std::vector<char> alive(n, 1);
  std::vector<int32_t> mark(n, -1), order, tmp;
  std::vector<std::vector<int32_t>> nb_at_elim(n);
  round_ptr.assign(1, 0);
  auto tail_deg = [&](int32_t i) { int d = 0; for (int32_t j : adj[i]) if (!hub[j]) d++; return d; };
  for (int r = 0; r < opt.max_rounds; r++) {
    std::vector<std::pair<int64_t, int32_t>> keyed;
    for (int32_t i = 0; i < n; i++)
      if (!hub[i] && alive[i]) {
        int td = tail_deg(i);
        if (td <= opt.max_tail_degree && (int)adj[i].size() <= opt.max_degree)
          keyed.push_back({(int64_t)td * 1000000 + (int64_t)adj[i].size(), i});
      }
    std::sort(keyed.begin(), keyed.end());
    std::vector<int32_t> I;
    for (auto& kv : keyed) {
      int32_t i = kv.second;
      bool ok = true;
      for (int32_t j : adj[i]) if (!hub[j] && mark[j] == r) { ok = false; break; }
      if (ok) { mark[i] = r; I.push_back(i); }
    }
    // stop when a round no longer pays for its two dependent kernel launches
    if (I.empty() || (r > 0 && (int)I.size() < opt.min_round)) { for (int32_t i : I) mark[i] = -1; break; }
    for (int32_t i : I) {
      auto& nb = adj[i];
      for (int32_t t : nb) {
        if (hub[t]) continue;
        tmp.clear();
        auto& a = adj[t];
        size_t x = 0, y = 0;
        while (x < a.size() || y < nb.size()) {
          int32_t v;
          if (y >= nb.size() || (x < a.size() && a[x] < nb[y])) v = a[x++];
          else if (x >= a.size() || nb[y] < a[x]) v = nb[y++];
          else { v = a[x]; x++; y++; }
          if (v != i && v != t) tmp.push_back(v);
        }
        a.swap(tmp);
      }
      alive[i] = 0;
      nb_at_elim[i] = nb;
      order.push_back(i);
    }
    round_ptr.push_back((int32_t)order.size());
  }
  nrounds = (int32_t)round_ptr.size() - 1;
  ns = (int32_t)order.size();
  m = n - ns;
  mpad = (int32_t)(ceil_div(std::max(m, 1), 64) * 64);
  if (m == 0) mpad = 0;
  perm = order;
  for (int32_t i = 0; i < n; i++) if (alive[i]) perm.push_back(i);
  iperm.assign(n, -1);
  for (int32_t q = 0; q < n; q++) iperm[perm[q]] = q;

  lap("elimination rounds");
  // ---- sparse structure in new indices
  ent_ptr.assign(ns + 1, 0);
  std::vector<int32_t> nbr;
  std::vector<int32_t> ent_piv;
  for (int32_t p = 0; p < ns; p++) {
    std::vector<int32_t> v;
    for (int32_t o : nb_at_elim[perm[p]]) v.push_back(iperm[o]);
    std::sort(v.begin(), v.end());
    for (int32_t q : v) {
      if (q <= p) throw KinError(ERR_DEVICE, "internal: elimination order violated");
      nbr.push_back(q); ent_piv.push_back(p);
    }
    ent_ptr[p + 1] = (int32_t)nbr.size();
  }
  nnzU = (int64_t)nbr.size();
  auto align = [](int64_t x) { return (x + 7) / 8 * 8; };
  off_diag = 0;
  off_U = align(ns);
  off_L = align(off_U + nnzU);
  off_S = align(off_L + nnzU);
  off_y = align(off_S + (int64_t)mpad * mpad);
  off_x = align(off_y + n);
  // the vectors of a solve - right-hand side / solution y, x and (explicit triangular inverses) y1, t - sit in ONE
  // contiguous window [off_y, off_vec_end): the resident integrator keeps that window in LDS
  off_y1 = align(off_x + mpad + 8);
  off_t = align(off_y1 + ns);
  off_vec_end = align(off_t + ns + 8);
  int64_t w_end = off_vec_end;

  // ---- in the background from here on (they need the elimination's structure and the offsets above, nothing later): the
  // scatter map of J and the Schur update plans of the rounds - 0.05 of the analysis' 0.20 s at 10k species, 0.49 of 1.39 s at
  // 50k, next to the monomials, the symbolic LZ / NVU products and the fused plans on this thread
  auto pos_of = [&](int32_t j, int32_t c) -> int64_t {  // location of W[j][c] (new indices), j,c later than the pivot
    if (j == c) return j < ns ? off_diag + j : off_S + (int64_t)(j - ns) * mpad + (j - ns);
    if (j >= ns && c >= ns) return off_S + (int64_t)(j - ns) * mpad + (c - ns);
    if (j < c) return off_U + find_entry(ent_ptr, nbr, j, c);   // U row j
    return off_L + find_entry(ent_ptr, nbr, c, j);              // L column c
  };

  std::vector<int32_t> bg_jm, bg_yl, bg_xl;
  std::vector<SegPlanHost> bg_hp(nrounds);
  std::vector<int64_t> bg_macs(nrounds, 0);
  std::exception_ptr bg_err;
  std::thread bg_thread([&] {
    try {
      bg_jm.resize(nnzJ);
      for (int32_t i = 0; i < n; i++)
        for (int32_t e = j_ptr[i]; e < j_ptr[i + 1]; e++) {
          int32_t j = j_col[e];
          int64_t p = pos_of(iperm[i], iperm[j]);
          bg_jm[e] = (int32_t)p | (i == j ? (int32_t)0x80000000 : 0);
        }
      bg_yl.resize(n); bg_xl.resize(n);
      for (int32_t v = 0; v < n; v++) {
        bg_yl[v] = (int32_t)(off_y + iperm[v]);
        bg_xl[v] = iperm[v] < ns ? (int32_t)(off_y + iperm[v]) : (int32_t)(off_x + iperm[v] - ns);
      }
      // one host thread per round: the triple lists (1 M entries at 10k species) are sorted and laid out independently
      std::vector<std::exception_ptr> errs(nrounds);
      std::vector<std::thread> th;
      for (int r = 0; r < nrounds; r++)
        th.emplace_back([&, r] {
          try {
            std::vector<Triple> tr;
            for (int32_t p = round_ptr[r]; p < round_ptr[r + 1]; p++)
              for (int32_t ej = ent_ptr[p]; ej < ent_ptr[p + 1]; ej++)
                for (int32_t ec = ent_ptr[p]; ec < ent_ptr[p + 1]; ec++)
                  tr.push_back({pos_of(nbr[ej], nbr[ec]), (int32_t)(off_L + ej), (int32_t)(off_U + ec)});
            bg_macs[r] = (int64_t)tr.size();
            bg_hp[r] = plan_from_triples(tr, false, 0, 0);
          } catch (...) { errs[r] = std::current_exception(); }
        });
      for (auto& t : th) t.join();
      for (auto& e : errs) if (e && !bg_err) bg_err = e;
    } catch (...) { bg_err = std::current_exception(); }
  });
  struct Joiner { std::thread& t; void join() { if (t.joinable()) t.join(); } ~Joiner() { join(); } } bg_join{bg_thread};

  // ---- explicit inverses of the sparse triangular blocks (symbolic): monomials along the elimination DAG
  struct Mono { float sign; std::vector<int32_t> fac; };
  std::vector<int32_t> m_ent_ptr{0}, m_ptr{0}, m_fac, m_dst;
  std::vector<float> m_sign;
  std::vector<std::vector<std::pair<int32_t, int32_t>>> z_cols(ns), v_cols(ns);   // per row: (column, entry slot)
  explicit_tri = !(getenv("KIN_LU_EXPLICIT") && atoi(getenv("KIN_LU_EXPLICIT")) == 0) && ns > 0;
  if (explicit_tri) {
    // rows of L11 and U11 (entries between two sparse pivots)
    std::vector<std::vector<std::pair<int32_t, int32_t>>> Lrow(ns), Urow(ns);   // (other index, entry e)
    for (int32_t p = 0; p < ns; p++)
      for (int32_t e = ent_ptr[p]; e < ent_ptr[p + 1]; e++)
        if (nbr[e] < ns) { Lrow[nbr[e]].push_back({p, e}); Urow[p].push_back({nbr[e], e}); }
    const int64_t mono_limit = 4000000;
    int64_t total = 0;
    off_dinv = w_end;                       // provisional: positions needed as factor indices below
    const int64_t dinv0 = off_dinv;
    // Z = L11^-1: Z[i,:] = e_i - sum_k L[i,k] Z[k,:], rows in increasing order
    std::vector<std::map<int32_t, std::vector<Mono>>> zr(ns), vr(ns);
    for (int32_t i = 0; i < ns && explicit_tri; i++)
      for (auto& ke : Lrow[i]) {
        if (!explicit_tri) break;
        const int32_t k = ke.first, le = (int32_t)(off_L + ke.second);
        zr[i][k].push_back(Mono{-1.0f, {le}});
        total++;
        for (auto& pm : zr[k])
          for (const Mono& mo : pm.second) {
            if (!explicit_tri) break;
            Mono nm{-mo.sign, mo.fac};
            nm.fac.push_back(le);
            zr[i][pm.first].push_back(std::move(nm));
            if (++total > mono_limit) { explicit_tri = false; break; }
          }
      }
    // V' = (I + D^-1 U11s)^-1: V'[i,:] = e_i - sum_c (U[i,c] dinv_i) V'[c,:], rows in decreasing order; V = V' D^-1
    for (int32_t i = ns - 1; i >= 0 && explicit_tri; i--)
      for (auto& ce : Urow[i]) {
        if (!explicit_tri) break;
        const int32_t c = ce.first, ue = (int32_t)(off_U + ce.second), di = (int32_t)(dinv0 + i);
        vr[i][c].push_back(Mono{-1.0f, {ue, di}});
        total++;
        for (auto& jm : vr[c])
          for (const Mono& mo : jm.second) {
            if (!explicit_tri) break;
            Mono nm{-mo.sign, mo.fac};
            nm.fac.push_back(ue); nm.fac.push_back(di);
            vr[i][jm.first].push_back(std::move(nm));
            if (++total > mono_limit) { explicit_tri = false; break; }
          }
      }
    if (explicit_tri) {
      // entry slots: Z off-diagonals (stored NEGATED: y1_i = b_i - sum Z'[i,p] b_p), then V including its diagonal
      nnzZ = 0; nnzV = 0;
      for (int32_t i = 0; i < ns; i++) nnzZ += (int64_t)zr[i].size();
      for (int32_t i = 0; i < ns; i++) nnzV += (int64_t)vr[i].size() + 1;
      off_dinv = w_end;
      off_Z = align(off_dinv + ns);
      off_V = align(off_Z + nnzZ);
      w_end = align(off_V + nnzV + 8);
      int64_t ez = 0, ev = 0;
      auto emit = [&](int64_t dst_pos, const std::vector<Mono>& monos, float flip, int32_t extra) {
        for (const Mono& mo : monos) {
          m_sign.push_back(flip * mo.sign);
          for (int32_t f : mo.fac) m_fac.push_back(f);
          if (extra >= 0) m_fac.push_back(extra);
          m_ptr.push_back((int32_t)m_fac.size());
        }
        m_dst.push_back((int32_t)dst_pos);
        m_ent_ptr.push_back((int32_t)m_sign.size());
      };
      for (int32_t i = 0; i < ns; i++)
        for (auto& pm : zr[i]) { z_cols[i].push_back({pm.first, (int32_t)ez}); emit(off_Z + ez, pm.second, -1.0f, -1); ez++; }
      for (int32_t i = 0; i < ns; i++) {
        v_cols[i].push_back({i, (int32_t)ev});
        emit(off_V + ev, std::vector<Mono>{Mono{1.0f, {}}}, 1.0f, (int32_t)(off_dinv + i));   // V[i,i] = 1 / d_i
        ev++;
        for (auto& jm : vr[i]) { v_cols[i].push_back({jm.first, (int32_t)ev}); emit(off_V + ev, jm.second, 1.0f, (int32_t)(off_dinv + jm.first)); ev++; }
      }
      n_monomials = (int64_t)m_sign.size();
      n_mono_ent = (int32_t)m_dst.size();
    }
  }
  lap("monomials of Z and V");
  // ---- fused solve (symbolic): LZ = L21 * L11^-1 and NVU = -(U11^-1 * U12), so that a solve is three dependent
  // launches: [y1 = Z b1 ; y2 = b2 - LZ b1] | x2 = S^-1 y2 | x1 = V y1 + NVU x2
  struct Prod { int32_t a, b; };
  // rows of LZ / NVU as flat CSR: per entry its column, the position of its direct term (LZ: L21[j,p], else -1) and its
  // products [pptr[e], pptr[e + 1]). Built row by row from a small sorted scratch list (no per-row maps: the symbolic
  // products are 1.7 M terms at 10k species, 13 M at 50k, and the analysis is part of every first solve of a network).
  struct FusedRows {
    std::vector<int64_t> row_ptr{0}, pptr{0};
    std::vector<int32_t> col, direct;
    std::vector<Prod> prod;
    int64_t n_entries() const { return (int64_t)col.size(); }
  };
  struct Item { int32_t col, a, b; };      // b < 0: direct term at position a
  auto flush_row = [](std::vector<Item>& tmp, FusedRows& F) {
    std::sort(tmp.begin(), tmp.end(), [](const Item& x, const Item& y) { return x.col != y.col ? x.col < y.col : x.b < y.b; });
    for (size_t q = 0; q < tmp.size();) {
      const int32_t c = tmp[q].col;
      int32_t dir = -1;
      for (; q < tmp.size() && tmp[q].col == c; q++) {
        if (tmp[q].b < 0) dir = tmp[q].a; else F.prod.push_back({tmp[q].a, tmp[q].b});
      }
      F.col.push_back(c); F.direct.push_back(dir); F.pptr.push_back((int64_t)F.prod.size());
    }
    F.row_ptr.push_back(F.n_entries());
    tmp.clear();
  };
  FusedRows LZ, NVU;
  fused_tri = explicit_tri && m > 0 && !(getenv("KIN_LU_FUSED") && atoi(getenv("KIN_LU_FUSED")) == 0);
  if (fused_tri) {
    const int64_t prod_limit = 16000000;
    // L21 entries bucketed by dense row
    std::vector<int64_t> dptr(m + 1, 0);
    for (int32_t e = 0; e < (int32_t)nbr.size(); e++) if (nbr[e] >= ns) dptr[nbr[e] - ns + 1]++;
    for (int32_t j = 0; j < m; j++) dptr[j + 1] += dptr[j];
    std::vector<int32_t> dk(dptr[m]), de(dptr[m]);
    {
      std::vector<int64_t> fill(dptr.begin(), dptr.end() - 1);
      for (int32_t k = 0; k < ns; k++)
        for (int32_t e = ent_ptr[k]; e < ent_ptr[k + 1]; e++)
          if (nbr[e] >= ns) { const int64_t q = fill[nbr[e] - ns]++; dk[q] = k; de[q] = e; }
    }
    // Rows are independent: each of the two products is built in `parts` row ranges of equal work on host threads of their
    // own and the pieces are concatenated (0.036 -> 0.011 s at 10k species, 0.30 -> 0.08 s at 50k on the GPU box's 16 cores)
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(6, (int64_t)std::thread::hardware_concurrency() / 2));
    auto lz_rows = [&](int32_t j0, int32_t j1, FusedRows& F) {
      std::vector<Item> tmp;
      for (int32_t j = j0; j < j1; j++) {
        for (int64_t q = dptr[j]; q < dptr[j + 1]; q++) {
          const int32_t k = dk[q], le = (int32_t)(off_L + de[q]);
          tmp.push_back({k, le, -1});
          for (auto& ce : z_cols[k]) tmp.push_back({ce.first, le, (int32_t)(off_Z + ce.second)});     // Z'[k, ce.first], ce.first < k
        }
        flush_row(tmp, F);
        if ((int64_t)F.prod.size() > prod_limit) return;       // the caller sees the overflow in the total
      }
    };
    auto nvu_rows = [&](int32_t i0, int32_t i1, FusedRows& F) {
      std::vector<Item> tmp;
      for (int32_t i = i0; i < i1; i++) {
        for (auto& je : v_cols[i]) {                   // V[i, je.first], je.first >= i
          const int32_t j = je.first;
          for (int32_t e = ent_ptr[j]; e < ent_ptr[j + 1]; e++)
            if (nbr[e] >= ns) tmp.push_back({nbr[e] - ns, (int32_t)(off_V + je.second), (int32_t)(off_U + e)});
        }
        flush_row(tmp, F);
        if ((int64_t)F.prod.size() > prod_limit) return;
      }
    };
    auto concat = [](std::vector<FusedRows>& P, FusedRows& F, int64_t rows_expected) {
      size_t ne = 0, np = 0, nr = 0;
      for (FusedRows& q : P) { ne += q.col.size(); np += q.prod.size(); nr += q.row_ptr.size() - 1; }
      F.row_ptr.reserve(nr + 1); F.pptr.reserve(ne + 1); F.col.reserve(ne); F.direct.reserve(ne); F.prod.reserve(np);
      for (FusedRows& q : P) {
        const int64_t e0 = F.n_entries(), p0 = (int64_t)F.prod.size();
        for (size_t r = 1; r < q.row_ptr.size(); r++) F.row_ptr.push_back(e0 + q.row_ptr[r]);
        for (size_t e = 1; e < q.pptr.size(); e++) F.pptr.push_back(p0 + q.pptr[e]);
        F.col.insert(F.col.end(), q.col.begin(), q.col.end());
        F.direct.insert(F.direct.end(), q.direct.begin(), q.direct.end());
        F.prod.insert(F.prod.end(), q.prod.begin(), q.prod.end());
        q = FusedRows{};
      }
      return (int64_t)F.row_ptr.size() - 1 == rows_expected;      // false: a part stopped at the product limit
    };
    std::vector<FusedRows> lzp(parts), nvup(parts);
    lap("  (buckets)");
    {
      // row ranges of equal WORK (the dense rows of the hubs carry most of LZ)
      std::vector<int64_t> wl(m + 1, 0), wn(ns + 1, 0), dense_of(ns, 0);
      for (int32_t k = 0; k < ns; k++)
        for (int32_t e = ent_ptr[k]; e < ent_ptr[k + 1]; e++) dense_of[k] += nbr[e] >= ns;
      for (int32_t j = 0; j < m; j++) {
        int64_t w = 1;
        for (int64_t q = dptr[j]; q < dptr[j + 1]; q++) w += 1 + (int64_t)z_cols[dk[q]].size();
        wl[j + 1] = wl[j] + w;
      }
      for (int32_t i = 0; i < ns; i++) {
        int64_t w = 1;
        for (auto& je : v_cols[i]) w += dense_of[je.first];
        wn[i + 1] = wn[i] + w;
      }
      auto cut = [&](const std::vector<int64_t>& w, int q) {       // first row of part q
        if (q <= 0) return (int32_t)0;
        if (q >= parts) return (int32_t)(w.size() - 1);
        return (int32_t)(std::lower_bound(w.begin(), w.end(), w.back() * q / parts) - w.begin());
      };
      std::vector<std::thread> th;
      std::vector<std::exception_ptr> errs(2 * parts);
      for (int q = 0; q < parts; q++) {
        const int32_t j0 = std::min(cut(wl, q), m), j1 = std::min(cut(wl, q + 1), m), i0 = std::min(cut(wn, q), ns), i1 = std::min(cut(wn, q + 1), ns);
        th.emplace_back([&, q, j0, j1] { try { lz_rows(j0, j1, lzp[q]); } catch (...) { errs[q] = std::current_exception(); } });
        th.emplace_back([&, q, i0, i1] { try { nvu_rows(i0, i1, nvup[q]); } catch (...) { errs[parts + q] = std::current_exception(); } });
      }
      for (auto& t : th) t.join();
      for (auto& e : errs) if (e) std::rethrow_exception(e);
    }
    lap("  (symbolic product rows)");
    bool lz_ok = false, nvu_ok = false;
    { std::thread tc([&] { lz_ok = concat(lzp, LZ, m); }); nvu_ok = concat(nvup, NVU, ns); tc.join(); }
    lap("  (concatenation)");
    const int64_t nprod = (int64_t)LZ.prod.size() + (int64_t)NVU.prod.size();
    if (!lz_ok || !nvu_ok || nprod > prod_limit) fused_tri = false;
    if (fused_tri) {
      nnzLZ = LZ.n_entries(); nnzNVU = NVU.n_entries();
      // LZ / NVU entries get IDs in a virtual range first; their storage is assigned below, in the order in which
      // the solve reads them (value-ordered plans)
      off_LZ = (int64_t)1 << 30;
      off_NVU = off_LZ + nnzLZ;
      if (off_NVU + nnzNVU >= ((int64_t)1 << 31)) throw KinError(ERR_UNSUPPORTED, "fused solve: too many entries");
      off_zero = w_end;
      w_end = align(off_zero + 8);
      n_fused_products = nprod;
    }
  }
  if (getenv("KIN_LU_DEBUG") && fused_tri) {
    auto hist = [](const char* name, const FusedRows& F) {
      long long h[6] = {0, 0, 0, 0, 0, 0}, mx = 0;
      for (size_t r = 0; r + 1 < F.row_ptr.size(); r++) {
        const long long l = (long long)(F.row_ptr[r + 1] - F.row_ptr[r]);
        mx = std::max(mx, l);
        h[l <= 8 ? 0 : l <= 64 ? 1 : l <= 256 ? 2 : l <= 1024 ? 3 : l <= 4096 ? 4 : 5]++;
      }
      fprintf(stderr, "[lu] %s row lengths: <=8:%lld <=64:%lld <=256:%lld <=1024:%lld <=4096:%lld more:%lld max:%lld\n", name, h[0], h[1], h[2], h[3], h[4], h[5], mx);
    };
    hist("LZ", LZ); hist("NVU", NVU);
  }
  if (getenv("KIN_LU_DEBUG"))
    fprintf(stderr, "[lu] n=%d ns=%d m=%d rounds=%d nnzU=%lld nnzZ=%lld nnzV=%lld monomials=%lld explicit=%d fused=%d nnzLZ=%lld nnzNVU=%lld products=%lld\n",
            n, ns, m, nrounds, (long long)nnzU, (long long)nnzZ, (long long)nnzV, (long long)n_monomials, (int)explicit_tri, (int)fused_tri,
            (long long)nnzLZ, (long long)nnzNVU, (long long)n_fused_products);
  lap("LZ / NVU symbolic products");
  // ---- J scatter map and per-round Schur update plans: built in the background since the elimination (above), uploaded here
  bg_join.join();
  if (bg_err) std::rethrow_exception(bg_err);
  up(jmap, bg_jm, s);
  up(ent_pivot, ent_piv, s);
  up(yloc, bg_yl, s); up(xloc, bg_xl, s);
  sync(s);
  schur.clear(); fwd.clear(); bwd.clear();
  schur.resize(nrounds); fwd.resize(nrounds); bwd.resize(nrounds);
  schur_macs = 0;
  for (int r = 0; r < nrounds; r++) { schur_macs += bg_macs[r]; up(schur[r], bg_hp[r], s); }
  { std::vector<SegPlanHost>().swap(bg_hp); }
  lap("scatter map + Schur plans");
  // ---- forward substitution: y_q -= sum_{p < q, q in nb(p)} L[q][p] * y_p, grouped by the round of q
  {
    std::vector<std::vector<Triple>> per_round(nrounds + 1);
    auto round_of = [&](int32_t q) {
      if (q >= ns) return nrounds;
      return (int)(std::upper_bound(round_ptr.begin(), round_ptr.end(), q) - round_ptr.begin()) - 1;
    };
    for (int32_t p = 0; p < ns; p++)
      for (int32_t e = ent_ptr[p]; e < ent_ptr[p + 1]; e++) {
        int32_t q = nbr[e];
        per_round[round_of(q)].push_back({off_y + q, (int32_t)(off_L + e), (int32_t)(off_y + p)});
      }
    for (int r = 0; r < nrounds; r++) up(fwd[r], plan_from_triples(per_round[r], false, 0, 0), s);
    if (explicit_tri)   // the dense rows read y1 from its own vector (the sparse rows are not updated in place any more)
      for (Triple& t : per_round[nrounds]) t.b = (int32_t)(off_y1 + (t.b - off_y));
    up(fwd_dense, plan_from_triples(per_round[nrounds], false, 0, 0), s);
  }
  if (explicit_tri && !fused_tri) {
    // y1_i = b_i - sum_p Z'[i,p] b_p
    {
      std::vector<int32_t> ptr{0}, dst, aux, a, b;
      for (int32_t i = 0; i < ns; i++) {
        for (auto& ce : z_cols[i]) { a.push_back((int32_t)(off_Z + ce.second)); b.push_back((int32_t)(off_y + ce.first)); }
        ptr.push_back((int32_t)a.size()); dst.push_back((int32_t)(off_y1 + i)); aux.push_back((int32_t)(off_y + i));
      }
      a.push_back(0); b.push_back(0);
      up(fwdZ, build_seg_plan(ns, ptr.data(), dst.data(), a.data(), b.data(), nullptr, false, aux.data()), s);
    }
    // t_p = y1_p - sum_{c dense} U[p,c] x2_c
    {
      std::vector<int32_t> ptr{0}, dst, aux, a, b;
      for (int32_t p = 0; p < ns; p++) {
        for (int32_t e = ent_ptr[p]; e < ent_ptr[p + 1]; e++)
          if (nbr[e] >= ns) { a.push_back((int32_t)(off_U + e)); b.push_back((int32_t)(off_x + nbr[e] - ns)); }
        ptr.push_back((int32_t)a.size()); dst.push_back((int32_t)(off_t + p)); aux.push_back((int32_t)(off_y1 + p));
      }
      a.push_back(0); b.push_back(0);
      up(bwdT, build_seg_plan(ns, ptr.data(), dst.data(), a.data(), b.data(), nullptr, false, aux.data()), s);
    }
    // x1_i = sum_j V[i,j] t_j
    {
      std::vector<int32_t> ptr{0}, dst, a, b;
      for (int32_t i = 0; i < ns; i++) {
        for (auto& ce : v_cols[i]) { a.push_back((int32_t)(off_V + ce.second)); b.push_back((int32_t)(off_t + ce.first)); }
        ptr.push_back((int32_t)a.size()); dst.push_back((int32_t)(off_y + i));
      }
      up(bwdV, build_seg_plan(ns, ptr.data(), dst.data(), a.data(), b.data(), nullptr, false), s);
    }
  }
  if (fused_tri) {
    // The two gather stages of the solve read each factor value exactly once, so the values are STORED in the order the
    // stages read them (value-ordered plans: no index array for the first factor, coalesced value stream). Step 1: the
    // stage structures with value IDs; step 2: storage positions from the plans' payload slots; step 3: every producer
    // (monomial kernel for Z and V, the LZ / NVU product launches) writes to the relocated positions.
    std::vector<int32_t> idA, idC, slotA, slotC;
    SegPlanHost PA, PC;
    auto build_A = [&] {   // stage A: y1_i = b_i - sum_p Z'[i,p] b_p (sparse rows) ; y2_j = b2_j - sum_p LZ[j,p] b_p (dense rows, in place)
      std::vector<int32_t> ptr{0}, dst, aux, b;
      for (int32_t i = 0; i < ns; i++) {
        for (auto& ce : z_cols[i]) { idA.push_back((int32_t)(off_Z + ce.second)); b.push_back((int32_t)(off_y + ce.first)); }
        ptr.push_back((int32_t)b.size()); dst.push_back((int32_t)(off_y1 + i)); aux.push_back((int32_t)(off_y + i));
      }
      for (int32_t j = 0; j < m; j++) {
        for (int64_t en = LZ.row_ptr[j]; en < LZ.row_ptr[j + 1]; en++) { idA.push_back((int32_t)(off_LZ + en)); b.push_back((int32_t)(off_y + LZ.col[en])); }
        ptr.push_back((int32_t)b.size()); dst.push_back((int32_t)(off_y + ns + j)); aux.push_back((int32_t)(off_y + ns + j));
      }
      b.push_back(0);
      PA = build_seg_plan((int64_t)ns + m, ptr.data(), dst.data(), nullptr, b.data(), nullptr, false, aux.data(), &slotA);
    };
    auto build_C = [&] {   // stage C: x1_i = sum_j V[i,j] y1_j + sum_c NVU[i,c] x2_c
      // (aux = the species behind the row: SEG_PROD_SET does not read it, the launch that fuses this stage with the
      // corrector update does - solver_kernels.hip: stagec_newton_kernel)
      std::vector<int32_t> ptr{0}, dst, aux, b;
      for (int32_t i = 0; i < ns; i++) {
        for (auto& ce : v_cols[i]) { idC.push_back((int32_t)(off_V + ce.second)); b.push_back((int32_t)(off_y1 + ce.first)); }
        for (int64_t en = NVU.row_ptr[i]; en < NVU.row_ptr[i + 1]; en++) { idC.push_back((int32_t)(off_NVU + en)); b.push_back((int32_t)(off_x + NVU.col[en])); }
        ptr.push_back((int32_t)b.size()); dst.push_back((int32_t)(off_y + i)); aux.push_back(perm[i]);
      }
      b.push_back(0);
      PC = build_seg_plan(ns, ptr.data(), dst.data(), nullptr, b.data(), nullptr, false, aux.data(), &slotC);
    };
    lap("  (forward plans)");
    { std::thread tA(build_A); build_C(); tA.join(); }
    lap("  (stage A | stage C plans)");
    off_VA = w_end;
    off_VC = align(off_VA + PA.ell_total + PA.long_total);
    w_end = align(off_VC + PC.ell_total + PC.long_total + 8);
    if (w_end >= ((int64_t)1 << 30)) throw KinError(ERR_UNSUPPORTED, "Newton matrix workspace exceeds int32 indexing");
    PA.val_base = (int32_t)off_VA; PC.val_base = (int32_t)off_VC;
    up(stageA, PA, s); up(stageC, PC, s);
    {
      const std::vector<int32_t> x2(perm.begin() + ns, perm.end());
      up(x2_species, x2, s);
      sync(s);   // the host vector dies here
    }
    // relocation of the value IDs
    std::vector<int32_t> relZ(nnzZ, -1), relV(nnzV, -1), relLZ(nnzLZ, -1), relNVU(nnzNVU, -1);
    auto place = [&](int32_t id, int32_t pos) {
      if (id >= off_NVU) relNVU[id - off_NVU] = pos;
      else if (id >= off_LZ) relLZ[id - off_LZ] = pos;
      else if (id >= off_V) relV[id - off_V] = pos;
      else relZ[id - off_Z] = pos;
    };
    for (size_t e = 0; e < idA.size(); e++) place(idA[e], (int32_t)(off_VA + slotA[e]));
    for (size_t e = 0; e < idC.size(); e++) place(idC[e], (int32_t)(off_VC + slotC[e]));
    auto R = [&](int32_t pos) -> int32_t {
      int32_t r = pos;
      if (pos >= off_NVU) r = relNVU[pos - off_NVU];
      else if (pos >= off_LZ) r = relLZ[pos - off_LZ];
      else if (pos >= off_V && pos < off_V + nnzV) r = relV[pos - off_V];
      else if (pos >= off_Z && pos < off_Z + nnzZ) r = relZ[pos - off_Z];
      if (r < 0) throw KinError(ERR_DEVICE, "internal: fused solve value without a storage position");
      return r;
    };
    for (int32_t& d : m_dst) d = R(d);
    lap("  (uploads + relocation)");
    // numeric products of a factorisation: LZ[j,p] = L21[j,p] - sum_k L21[j,k] Z'[k,p] ; NVU[i,c] = - sum_j V[i,j] U12[j,c]
    SegPlanHost Plz, Pnvu;
    auto build_lzp = [&] {
      std::vector<int32_t> ptr{0}, dst, aux, a, b;
      a.reserve(LZ.prod.size() + 1); b.reserve(LZ.prod.size() + 1);
      for (int64_t en = 0; en < LZ.n_entries(); en++) {
        for (int64_t q = LZ.pptr[en]; q < LZ.pptr[en + 1]; q++) { a.push_back(LZ.prod[q].a); b.push_back(R(LZ.prod[q].b)); }
        ptr.push_back((int32_t)a.size());
        dst.push_back(R((int32_t)(off_LZ + en)));
        aux.push_back(LZ.direct[en] >= 0 ? LZ.direct[en] : (int32_t)off_zero);
      }
      a.push_back(0); b.push_back(0);
      Plz = build_seg_plan((int64_t)dst.size(), ptr.data(), dst.data(), a.data(), b.data(), nullptr, false, aux.data());
    };
    auto build_nvup = [&] {
      std::vector<int32_t> ptr{0}, dst, a, b;
      a.reserve(NVU.prod.size() + 1); b.reserve(NVU.prod.size() + 1);
      for (int64_t en = 0; en < NVU.n_entries(); en++) {
        for (int64_t q = NVU.pptr[en]; q < NVU.pptr[en + 1]; q++) { a.push_back(R(NVU.prod[q].a)); b.push_back(NVU.prod[q].b); }
        ptr.push_back((int32_t)a.size());
        dst.push_back(R((int32_t)(off_NVU + en)));
      }
      a.push_back(0); b.push_back(0);
      Pnvu = build_seg_plan((int64_t)dst.size(), ptr.data(), dst.data(), a.data(), b.data(), nullptr, false);
    };
    { std::thread tL(build_lzp); build_nvup(); tL.join(); }
    lap("  (LZ | NVU product plans)");
    up(lz_build, Plz, s); up(nvu_build, Pnvu, s);
    sync(s);
  }
  if (explicit_tri) {
    up(mono_ent_ptr, m_ent_ptr, s); up(mono_ptr, m_ptr, s); up(mono_fac, m_fac, s); up(mono_dst, m_dst, s);
    up(mono_sign, m_sign, s);
    sync(s);
  }
  w_size = w_end;
  if (w_size >= (1ll << 31)) throw KinError(ERR_UNSUPPORTED, "Newton matrix workspace exceeds int32 indexing");
  lap("substitution + fused plans");
  // ---- backward substitution: x_p = (y_p - sum_c U[p][c] x_c) / diag_p
  for (int r = 0; r < nrounds; r++) {
    int32_t p0 = round_ptr[r], p1 = round_ptr[r + 1];
    std::vector<int32_t> ptr{0}, dst, aux, a, b;
    for (int32_t p = p0; p < p1; p++) {
      for (int32_t e = ent_ptr[p]; e < ent_ptr[p + 1]; e++) {
        int32_t c = nbr[e];
        a.push_back((int32_t)(off_U + e));
        b.push_back(c < ns ? (int32_t)(off_y + c) : (int32_t)(off_x + c - ns));
      }
      ptr.push_back((int32_t)a.size());
      dst.push_back((int32_t)(off_y + p));
      aux.push_back((int32_t)(off_diag + p));
    }
    a.push_back(0); b.push_back(0);  // keep data() valid for all-empty rounds
    up(bwd[r], build_seg_plan(p1 - p0, ptr.data(), dst.data(), a.data(), b.data(), nullptr, false, aux.data()), s);
  }

  if (!host_only) { pinv.alloc(2 * 32 * 32); slots.clear(); ensure_slots(1, s); }
  sync(s);
  lap("backward plans + first slot");
}

void SparseLU::alloc_slot(Slot& q, hipStream_t s) const {
  q.W.alloc((size_t)w_size);
  KIN_HIP(hipMemsetAsync(q.W.p, 0, (size_t)w_size * sizeof(double), s));
  q.S2.alloc((size_t)std::max(mpad, 64) * std::max(mpad, 64));
}

void SparseLU::ensure_slots(int nslots, hipStream_t s) {
  while ((int)slots.size() < nslots) {
    slots.emplace_back();
    alloc_slot(slots.back(), s);
  }
}

void SparseLU::factor_into(double c, const double* d_jvals, Slot& q, double* pinv_scratch, int* bad, hipStream_t s) {
  factor_sparse_into(c, d_jvals, q, bad, s);
  if (m > 0) q.sinv = launch_gauss_jordan(q.W.p + off_S, q.S2.p, mpad, pinv_scratch, bad, s);
}

// everything of a factorisation but the inverse of the dense Schur block (which is then complete in W + off_S, waiting to be
// inverted: by factor_into on the same stream, or - for several slots at once - by launch_gauss_jordan_batched, ensemble.cpp)
void SparseLU::factor_sparse_into(double c, const double* d_jvals, Slot& q, int* bad, hipStream_t s) {
  double* W = q.W.p;
  // zero everything up to the solve vectors, then scatter I - c*J
  KIN_HIP(hipMemsetAsync(W, 0, (size_t)off_y * sizeof(double), s));
  launch_lu_assemble(nnzJ, jmap.p, d_jvals, c, W, off_S, m, mpad, s);
  for (int r = 0; r < nrounds; r++) {
    launch_lu_scale(ent_ptr[round_ptr[r]], ent_ptr[round_ptr[r + 1]], ent_pivot.p, W, off_L, off_diag, bad, s);
    launch_segsum(schur[r].view(), SEG_PROD_SUB, W, W, SegExtra{}, s);
  }
  if (explicit_tri) {
    launch_lu_recip(ns, W + off_diag, W + off_dinv, s);
    launch_lu_mono(n_mono_ent, mono_ent_ptr.p, mono_ptr.p, mono_fac.p, mono_sign.p, mono_dst.p, W, s);
    if (fused_tri) {
      launch_segsum(lz_build.view(), SEG_PROD_AUXSUB, W, W, SegExtra{}, s);
      launch_segsum(nvu_build.view(), SEG_PROD_NEG, W, W, SegExtra{}, s);
    }
  }
  q.c_fact = c;
  q.crate = 1.0;
  q.valid = true;
}

void SparseLU::solve_newton(int slot, NewtonFuse f, hipStream_t s) {
  Slot& q = slots[slot];
  double* W = q.W.p;
  SegExtra ex;
  ex.skip = f.skip;
  launch_segsum(stageA.view(), SEG_PROD_AUXSUB, W, W, ex, s);
  launch_gemv(q.sinv, mpad, m, W + off_y + ns, W + off_x, f.skip, s);
  f.m = m; f.off_x = (int32_t)off_x; f.x2_species = x2_species.p;
  launch_stagec_newton(stageC.view(), W, f, s);
}

void SparseLU::solve(const int* skip, int slot, hipStream_t s) {
  Slot& q = slots[slot];
  double* W = q.W.p;
  SegExtra ex;
  ex.skip = skip;
  if (fused_tri) {
    launch_segsum(stageA.view(), SEG_PROD_AUXSUB, W, W, ex, s);
    launch_gemv(q.sinv, mpad, m, W + off_y + ns, W + off_x, skip, s);
    launch_segsum(stageC.view(), SEG_PROD_SET, W, W, ex, s);
    return;
  }
  if (explicit_tri) {
    launch_segsum(fwdZ.view(), SEG_PROD_AUXSUB, W, W, ex, s);
    if (m > 0) {
      launch_segsum(fwd_dense.view(), SEG_PROD_SUB, W, W, ex, s);
      launch_gemv(q.sinv, mpad, m, W + off_y + ns, W + off_x, skip, s);
    }
    launch_segsum(bwdT.view(), SEG_PROD_AUXSUB, W, W, ex, s);
    launch_segsum(bwdV.view(), SEG_PROD_SET, W, W, ex, s);
    return;
  }
  for (int r = 1; r < nrounds; r++) launch_segsum(fwd[r].view(), SEG_PROD_SUB, W, W, ex, s);
  if (m > 0) {
    if (ns > 0) launch_segsum(fwd_dense.view(), SEG_PROD_SUB, W, W, ex, s);
    launch_gemv(q.sinv, mpad, m, W + off_y + ns, W + off_x, skip, s);
  }
  for (int r = nrounds - 1; r >= 0; r--) launch_segsum(bwd[r].view(), SEG_PROD_SUB_DIV, W, W, ex, s);
}

}  // namespace kin
