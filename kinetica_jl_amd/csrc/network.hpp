// Host-side "network compiler": turns RxData's ragged stoichiometry (reference
// src/exploration/network.jl:193-203) into the fixed-width, data-driven tables the device
// kernels stream. Nothing here is generated per network beyond index tables - there is no
// symbolic build (the reference's Catalyst -> ModelingToolkit codegen, methods.jl:140-158,
// is replaced by table lookups).
#pragma once
#include <cstdint>
#include <vector>

#include "common.hpp"

namespace kin {

// Deterministic load-balanced gather-sum plan: out[dst(row)] = sum over the row's entries.
// Short rows are packed 64 to a wavefront in a transposed (ELL) layout so that lane reads are
// coalesced; medium rows (<= SEG_LEN entries) take one wavefront each, longer rows one whole
// workgroup each; every sum is formed in a fixed order (bitwise reproducible, no atomics, no
// partial sums in memory).
struct SegPlanHost {
  // Everything these plans drive is cache resident and latency bound, so the sizes minimise the
  // longest dependent chain per wavefront: ELL rows are walked 4 columns at a time with all loads
  // in flight, a segment is consumed in ONE pass of 4 entries per lane.
  static constexpr int SHORT_MAX = 8;    // rows up to this many entries go to the ELL groups
  static constexpr int SEG_LEN = 256;    // longest row one wavefront takes (4 entries per lane; the kernel also handles up to 1024 at 16 per lane,
                                         // measured slower than a workgroup per row: 0.650 vs 0.614 s on the C3 solve)
  static constexpr int BLK_PASS = 12288; // entries a 1024-thread workgroup consumes per pass of a long row (12 per thread)
  // ELL groups
  std::vector<int32_t> grp_off;  // G+1: first ELL column of each group
  std::vector<int32_t> grp_dst;  // G*64: output index per lane (-1 = idle lane)
  std::vector<int32_t> grp_aux;  // G*64: optional per-row auxiliary index (divisor position)
  std::vector<int32_t> ell_a, ell_b;
  std::vector<float> ell_c;      // 0.0f marks padding
  // medium rows (9 .. SEG_LEN entries): one wavefront each, payload copied contiguously (sorted by length, so that the
  // four / sixteen tasks of a workgroup are of a kind)
  std::vector<int32_t> seg_beg, seg_end;  // S
  std::vector<int32_t> seg_dst;           // S: output index
  std::vector<int32_t> seg_aux;           // S
  // long rows (> SEG_LEN entries): one whole workgroup each, BLK_PASS entries per pass, fixed-order reduction -
  // no partial sums in memory and no second (fix-up) launch behind the gather
  std::vector<int32_t> blk_beg, blk_end, blk_dst, blk_aux;   // B
  std::vector<int32_t> long_a, long_b;    // payload of segments and long rows
  std::vector<float> long_c;
  // Value-ordered product plans: when val_base >= 0 the first factor of payload slot q is src[val_base + q] (ELL slots
  // first, then the long payload) - no `a` index array, and the values are read as a coalesced stream instead of a
  // gather. The producers of those values write them at val_base + slot (build_seg_plan reports each entry's slot).
  int32_t val_base = -1;
  int32_t ell_total = 0, long_total = 0;     // payload slots of the ELL part / of the medium + long rows
  int32_t n_groups() const { return (int32_t)grp_off.size() - 1; }
  int32_t n_segs() const { return (int32_t)seg_beg.size(); }
  int32_t n_blks() const { return (int32_t)blk_beg.size(); }
};

// Build a plan from CSR rows. `dst[row]` is the output index of a row; rows with no entries
// are still scheduled (they produce 0) unless skip_empty. `b` and `c` may be null. With `b` the plan is a
// product plan (entries src[a] * src[b], no coefficients).
// `slot_of_entry` (optional, product plans): payload slot of every input entry (see SegPlanHost::val_base); with it
// given and `a` null the plan is built value-ordered (the caller sets val_base afterwards).
SegPlanHost build_seg_plan(int64_t n_rows, const int32_t* ptr, const int32_t* dst, const int32_t* a,
                           const int32_t* b, const float* c, bool skip_empty, const int32_t* aux = nullptr,
                           std::vector<int32_t>* slot_of_entry = nullptr);

struct NetworkHost {
  int64_t N = 0, R = 0;
  // per reaction: rate operands, rate = k * u[x0] * (x1 >= 0 ? u[x1] : 1)   (2A: x0 == x1)
  std::vector<int32_t> x0, x1;
  // per reaction: product instances expanded by stoichiometry (2C is listed as C, C); -1 = none. products_le2: every
  // reaction has at most two of them (the reference's max_molecularity = 2 holds for both sides, network.jl:275-279)
  std::vector<int32_t> y0, y1;
  bool products_le2 = true;
  // per reaction: up to 4 update slots (distinct species with non-zero net stoichiometry)
  std::vector<int32_t> slot_sp;   // 4*R, species id or -1
  std::vector<int32_t> slot_co;   // R, four signed bytes packed
  // reversible-pair records of the batched sweep: reaction `kf` and its exact reverse `kr`
  // (kr = -1: no partner) share one record: <= 4 species slots with the FORWARD net
  // stoichiometry; forward rate = k[kf] * prod_{coef<0} u^|coef|, reverse rate =
  // k[kr] * prod_{coef>0} u^|coef|, every slot receives coef * (forward - reverse).
  // Reactions with a species on both sides (colliders) stay unpaired, flagged by bit 31 of
  // pair_k[2p+1]'s companion word pair_ops (explicit operands).
  std::vector<uint32_t> pair_rec;  // 4 words per record: s01, s23, coefs, ops (explicit operands for unpaired records)
  std::vector<int32_t> pair_k;     // 2 per record: kf, kr
  std::vector<uint32_t> gen_rec8;   // 2 words per record: fixed-role 16-bit labels for the general LDS sweep (state fits LDS)
  std::vector<int32_t> gen_expl;    // records with explicit operands
  std::vector<int32_t> sweep_copy_species;   // species behind every extra accumulator entry (7 per split hub)
  std::vector<uint32_t> pair_rec64; // 2 words per record (four 14-bit labels with fixed roles, network.cpp); only when pairs_adjacent or pairs_block
  // Large-N sweep (state does not fit LDS): species are relabelled so that the `big_H` most
  // frequently referenced ones ("hubs") come first (each group kept in species-id order); records
  // carry labels. Hubs keep u and du in LDS; tail rates are gathered per tile of 2 * big_H labels from
  // `big_tail_ent` = (record, local label | coef << 24) pairs, sorted by record inside each tile.
  int32_t big_H = 0;
  bool big_tail_by_species = false;   // stream records address tail operands by SPECIES id (label = big_H + 64 + species):
                                      // operands come straight from the caller's u[b], no relabelled scratch copy of the tail
  std::vector<int32_t> big_spec_of_label;    // N
  std::vector<uint32_t> big_rec;             // 4 words per record, slots = labels (slow path of explicit-operand records)
  std::vector<uint32_t> big_rec8;            // 2 words per record: the stream format (kernels.hip: sweep_big_kernel)
  std::vector<int32_t> big_expl;             // records with explicit operands
  std::vector<int32_t> big_tail_ptr;         // tail tiles + 1 (entry offsets)
  std::vector<uint32_t> big_tail_ent;        // 2 words per entry
  bool pairs_adjacent = false;     // record p pairs reactions (2p, 2p+1): k streams as double2, no index load
  bool pairs_block = false;        // record p pairs reactions (p, P+p): forwards first, reverses behind (duplicate_reverse order)
  int64_t n_pairs() const { return (int64_t)pair_k.size() / 2; }
  // species-major CSR: du[i] = sum_e sp_coef[e] * rate[sp_rxn[e]]
  std::vector<int32_t> sp_ptr, sp_rxn;
  std::vector<float> sp_coef;
  // Jacobian: CSR pattern (sorted columns, diagonal present) and, per stored entry, the list of
  // contributions coef * drate[src], src = 2*r + w (w-th operand derivative of reaction r)
  std::vector<int32_t> j_ptr, j_col, j_diag;  // j_diag[i] = position of (i,i)
  std::vector<int32_t> jc_ptr, jc_src;
  std::vector<float> jc_coef;
  int64_t nnz() const { return (int64_t)j_col.size(); }
};

// Validates and compiles the flat ragged arrays (see kin_network_create in kinetica_hip.h).
// Throws KinError(ERR_INVALID_ARG | ERR_UNSUPPORTED).
NetworkHost compile_network(int64_t n_species, int64_t n_reactions, const int64_t* reac_ptr,
                            const int64_t* reac_idx, const int64_t* reac_sto, const int64_t* prod_ptr,
                            const int64_t* prod_idx, const int64_t* prod_sto, int index_base);

}  // namespace kin
