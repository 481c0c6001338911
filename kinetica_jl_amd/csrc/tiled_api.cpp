// extern "C" entry points of the tiled batched sweep (library order: tiled.hpp; kernels: tiled_kernels.hip).
#include "../../include/kinetica_hip.h"

#include <algorithm>
#include <cstdio>
#include <cmath>

#include "handle.hpp"
#include "tiled_kernels.hpp"

using namespace kin;

namespace {

// workgroup size: 10 staged doubles per thread must cover the LDS entries (tiled_kernels.hip)
int tiled_block_size(int64_t N) {
  return N <= 2300 ? 256 : (N <= 4850 ? 512 : 1024);
}

// builds and uploads the library order once per handle; throws ERR_UNSUPPORTED when the network is not tileable
void ensure_tiled(kin_network* h) {
  if (!h->tiled_tried) {
    h->tiled_tried = true;
    h->tiled = build_tiled(h->host, tiled_block_size(h->host.N), 0);
    const TiledHost& L = h->tiled;
    if (L.ok) {
      hipStream_t s = h->stream;
      std::vector<int32_t> rxn_of_slot((size_t)L.k_len(), -1);
      for (int32_t r = 0; r < L.R; r++) if (L.slot_of_reaction[r] >= 0) rxn_of_slot[L.slot_of_reaction[r]] = r;
      h->t_rec.upload(L.rec, s);
      h->t_copy.upload(L.copy_src, s);
      h->t_kf.upload(L.kf, s); h->t_kr.upload(L.kr, s); h->t_rxn_of_slot.upload(rxn_of_slot, s); h->t_kslot.upload(L.kslot, s);
      h->t_spec_of_lib.upload(L.species_of_lib, s); h->t_lib_of_spec.upload(L.lib_of_species, s);
      if (!L.identity) { h->t_stage_lib.upload(L.stage_lib, s); h->t_stage_off.upload(L.stage_off, s); }
      KIN_HIP(hipStreamSynchronize(s));
    }
  }
  if (!h->tiled.ok) throw KinError(ERR_UNSUPPORTED, "network has no tiled layout: " + h->tiled.why);
}

// Per-record Arrhenius parameters in library order: rebuilt once per kin_set_arrhenius, on the HANDLE's stream and waited
// for - consumers run on whatever stream the caller passes (the handle's own stream is non-blocking: nothing orders it
// against them), so the table must be complete before the first of them is enqueued; the device-wide wait in front also
// lets sweeps still reading the OLD table on other streams finish before it is overwritten.
void ensure_params(kin_network* h, hipStream_t) {
  if (!h->has_arrhenius) throw KinError(ERR_STATE, "Arrhenius parameters were never set");
  if (h->t_par_valid) return;
  const TiledHost& L = h->tiled;
  KIN_HIP(hipDeviceSynchronize());
  h->t_par.alloc((size_t)4 * std::max(1, L.P));
  launch_tiled_params(L.P, h->t_kf.p, h->t_kr.p, h->Ea.p, h->A.p, h->has_kmax, h->t_mult, h->t_par.p, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  h->t_par_valid = true;
}

TiledView view_of(kin_network* h) {
  const TiledHost& L = h->tiled;
  TiledView v{};
  v.KL = L.KL; v.has_singles = L.has_singles ? 1 : 0;
  v.N = L.N; v.P = L.P; v.h = L.h; v.n_copy = L.n_copy; v.wbase = L.wbase; v.E = L.E; v.T = L.T;
  v.win_cnt_max = 0;
  for (int32_t c : L.win_cnt) v.win_cnt_max = std::max(v.win_cnt_max, c);
  v.rec = (const uint2*)h->t_rec.p;
  for (int32_t t = 0; t < L.T; t++) {
    v.seginfo[t] = make_int4(L.seginfo[4 * t], L.seginfo[4 * t + 1], L.seginfo[4 * t + 2], L.seginfo[4 * t + 3]);
    v.segk[t] = L.seg_k[2 * t];
    v.win_off[t] = L.win_off[t]; v.win_cnt[t] = L.win_cnt[t];
  }
  v.copy_src = h->t_copy.p;
  v.par = (const double4*)h->t_par.p;
  v.has_kmax = h->has_kmax ? 1 : 0;
  v.inv_kmax = h->has_kmax ? 1.0 / h->k_max : 0.0;
  return v;
}

void require(bool c, int code, const char* msg) {
  if (!c) throw KinError(code, msg);
}

// caller's species order <-> library order for B states (only called when the two differ or the caller asks for a copy)
void states_convert(kin_network* h, int64_t B, bool to_lib, const double* d_in, double* d_out, hipStream_t s) {
  const int64_t N = h->host.N;
  if (h->tiled.identity) KIN_HIP(hipMemcpyAsync(d_out, d_in, (size_t)B * N * sizeof(double), hipMemcpyDeviceToDevice, s));
  else launch_permute_staged(N, B, to_lib, h->t_stage_lib.p, h->t_stage_off.p, d_in, d_out, s);
}

}  // namespace

#define KIN_TRY(h) try { KIN_HIP(hipSetDevice((h)->device));
#define KIN_CATCH(h)                                                        \
  }                                                                         \
  catch (const KinError& e) { (h)->err = e.what(); return e.code; }         \
  catch (const std::exception& e) { (h)->err = e.what(); return KIN_ERR_DEVICE; } \
  return KIN_OK;

extern "C" {

int kin_lib_layout(kin_network* h, int index_base, int64_t* k_len, int64_t* species_of_lib, int64_t* slot_of_reaction,
                   int32_t* species_identity, int64_t* info) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  ensure_tiled(h);
  const TiledHost& L = h->tiled;
  if (k_len) *k_len = L.k_len();
  if (species_of_lib) for (int32_t j = 0; j < L.N; j++) species_of_lib[j] = L.species_of_lib[j] + index_base;
  if (slot_of_reaction) for (int32_t r = 0; r < L.R; r++) slot_of_reaction[r] = L.slot_of_reaction[r] + index_base;
  if (species_identity) *species_identity = L.identity ? 1 : 0;
  if (info) { info[0] = L.h; info[1] = L.T; info[2] = L.P; info[3] = L.E; info[4] = L.n_copy; info[5] = L.BS; }
  KIN_CATCH(h)
}

int kin_lib_layout_host(int64_t n_species, int64_t n_reactions, const int64_t* reac_ptr, const int64_t* reac_idx,
                        const int64_t* reac_sto, const int64_t* prod_ptr, const int64_t* prod_idx, const int64_t* prod_sto,
                        int index_base, int hubs, int64_t* info, int64_t* species_of_lib, int64_t* slot_of_reaction,
                        uint64_t* rec, int32_t* rowtab, int32_t* seg_q, int32_t* win_off, int32_t* win_cnt, int32_t* copy_src,
                        int32_t* seg_k) {
  try {
    const NetworkHost H = compile_network(n_species, n_reactions, reac_ptr, reac_idx, reac_sto, prod_ptr, prod_idx, prod_sto, index_base);
    const TiledHost L = build_tiled(H, tiled_block_size(H.N), hubs);
    if (!L.ok) {
      if (getenv("KIN_TILED_DEBUG")) fprintf(stderr, "[tiled] no layout: %s\n", L.why.c_str());
      return KIN_ERR_UNSUPPORTED;
    }
    if (info) {
      info[0] = L.h; info[1] = L.T; info[2] = L.P; info[3] = L.E; info[4] = L.n_copy; info[5] = L.BS; info[6] = L.wbase;
      info[7] = L.seg_q.back(); info[8] = L.k_len(); info[9] = L.has_singles ? 1 : 0;
    }
    if (species_of_lib) for (int32_t j = 0; j < L.N; j++) species_of_lib[j] = L.species_of_lib[j];
    if (slot_of_reaction) for (int32_t r = 0; r < L.R; r++) slot_of_reaction[r] = L.slot_of_reaction[r];
    if (rec) for (int32_t p = 0; p < L.P; p++) rec[p] = (uint64_t)L.rec[2 * p] | ((uint64_t)L.rec[2 * p + 1] << 32);
    if (rowtab) std::copy(L.rowtab.begin(), L.rowtab.end(), rowtab);
    if (seg_q) std::copy(L.seg_q.begin(), L.seg_q.end(), seg_q);
    if (win_off) std::copy(L.win_off.begin(), L.win_off.end(), win_off);
    if (win_cnt) std::copy(L.win_cnt.begin(), L.win_cnt.end(), win_cnt);
    if (copy_src) std::copy(L.copy_src.begin(), L.copy_src.end(), copy_src);
    if (seg_k) std::copy(L.seg_k.begin(), L.seg_k.end(), seg_k);
  } catch (const KinError& e) {
    return e.code;
  } catch (const std::exception&) {
    return KIN_ERR_INVALID_ARG;
  }
  return KIN_OK;
}

int kin_states_to_lib_dev(kin_network* h, int64_t B, const double* d_in, double* d_out, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0 && d_in && d_out && d_in != d_out, ERR_INVALID_ARG, "bad arguments");
  ensure_tiled(h);
  states_convert(h, B, true, d_in, d_out, stream ? (hipStream_t)stream : h->stream);
  KIN_CATCH(h)
}

int kin_states_from_lib_dev(kin_network* h, int64_t B, const double* d_in, double* d_out, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0 && d_in && d_out && d_in != d_out, ERR_INVALID_ARG, "bad arguments");
  ensure_tiled(h);
  states_convert(h, B, false, d_in, d_out, stream ? (hipStream_t)stream : h->stream);
  KIN_CATCH(h)
}

int kin_rates_to_lib_dev(kin_network* h, int64_t B, const double* d_k, double* d_k_lib, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0 && d_k && d_k_lib && d_k != d_k_lib, ERR_INVALID_ARG, "bad arguments");
  ensure_tiled(h);
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const TiledHost& L = h->tiled;
  // record-wise (one 16-byte load per adjacent pair) when every record has two slots and the rows allow 16-byte accesses
  if (L.KL == 2 * (int64_t)L.P && h->host.R % 2 == 0 && ((((uintptr_t)d_k) | ((uintptr_t)d_k_lib)) & 15) == 0)
    launch_rates_to_lib_pairs(L.P, h->host.R, B, h->t_kf.p, h->t_kr.p, d_k, d_k_lib, s);
  else
    launch_gather_rows(L.k_len(), h->host.R, B, h->t_rxn_of_slot.p, d_k, d_k_lib, s);
  KIN_CATCH(h)
}

int kin_rate_table_lib_dev(kin_network* h, const double* T, int64_t n_stops, double* d_out) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(T != nullptr && d_out != nullptr && n_stops >= 0, ERR_INVALID_ARG, "bad arguments");
  ensure_tiled(h);
  ensure_params(h, h->stream);
  h->T_stops.upload(T, n_stops, h->stream);
  TiledPadSlots pads{};
  for (int32_t q : h->tiled.pad_slots) pads.slot[pads.n++] = q;
  launch_rate_table_lib(h->tiled.P, h->tiled.KL, n_stops, h->t_par.p, h->t_kslot.p, pads, h->has_kmax, h->k_max, h->T_stops.p, d_out, h->stream);
  KIN_HIP(hipStreamSynchronize(h->stream));
  KIN_CATCH(h)
}

int kin_rhs_tiled_dev(kin_network* h, int64_t B, const double* d_u_lib, const double* d_k_lib, const double* d_T,
                      double* d_du_lib, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0 && d_u_lib && d_du_lib, ERR_INVALID_ARG, "bad arguments");
  require((d_k_lib != nullptr) != (d_T != nullptr), ERR_INVALID_ARG, "exactly one of d_k_lib and d_T must be given");
  require((((uintptr_t)d_k_lib) & 15) == 0, ERR_INVALID_ARG, "d_k_lib must be 16-byte aligned");
  ensure_tiled(h);
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  if (d_T) ensure_params(h, s);
  if (h->tiled.P == 0) {   // a network without reactions: du = 0
    KIN_HIP(hipMemsetAsync(d_du_lib, 0, (size_t)B * h->host.N * sizeof(double), s));
    return KIN_OK;
  }
  launch_tiled_sweep(view_of(h), h->tiled.BS, h->n_cu, B, d_u_lib, d_k_lib, d_T, d_du_lib, s);
  KIN_CATCH(h)
}

int kin_rhs_batched_T_dev(kin_network* h, int64_t B, const double* d_u, const double* d_T, double* d_du, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0 && d_u && d_T && d_du, ERR_INVALID_ARG, "bad arguments");
  ensure_tiled(h);
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  ensure_params(h, s);
  const int64_t N = h->host.N;
  if (h->tiled.P == 0) {
    KIN_HIP(hipMemsetAsync(d_du, 0, (size_t)B * N * sizeof(double), s));
    return KIN_OK;
  }
  if (h->tiled.identity) {
    launch_tiled_sweep(view_of(h), h->tiled.BS, h->n_cu, B, d_u, nullptr, d_T, d_du, s);
  } else {
    // caller's species order: two layout conversions around the sweep (16 N bytes per state each way; a caller that
    // keeps its states in library order calls kin_rhs_tiled_dev and pays neither)
    h->t_u.alloc((size_t)B * N); h->t_du.alloc((size_t)B * N);
    states_convert(h, B, true, d_u, h->t_u.p, s);
    launch_tiled_sweep(view_of(h), h->tiled.BS, h->n_cu, B, h->t_u.p, nullptr, d_T, h->t_du.p, s);
    states_convert(h, B, false, h->t_du.p, d_du, s);
  }
  KIN_CATCH(h)
}

int kin_rhs_batched_klib_dev(kin_network* h, int64_t B, const double* d_u, const double* d_k_lib, double* d_du, void* stream) {
  if (!h) return KIN_ERR_INVALID_ARG;
  KIN_TRY(h)
  require(B > 0 && d_u && d_k_lib && d_du, ERR_INVALID_ARG, "bad arguments");
  require((((uintptr_t)d_k_lib) & 15) == 0, ERR_INVALID_ARG, "d_k_lib must be 16-byte aligned");
  ensure_tiled(h);
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const int64_t N = h->host.N;
  if (h->tiled.P == 0) {
    KIN_HIP(hipMemsetAsync(d_du, 0, (size_t)B * N * sizeof(double), s));
    return KIN_OK;
  }
  if (h->tiled.identity) {
    launch_tiled_sweep(view_of(h), h->tiled.BS, h->n_cu, B, d_u, d_k_lib, nullptr, d_du, s);
  } else {
    // the species permutation on the way in and out, each ONE coalesced pass through LDS (16 N bytes per state each way, next to
    // the sweep's 8 k_len + 16 N): the workspace rows live on the handle
    h->t_u.alloc((size_t)B * N); h->t_du.alloc((size_t)B * N);
    states_convert(h, B, true, d_u, h->t_u.p, s);
    launch_tiled_sweep(view_of(h), h->tiled.BS, h->n_cu, B, h->t_u.p, d_k_lib, nullptr, h->t_du.p, s);
    states_convert(h, B, false, h->t_du.p, d_du, s);
  }
  KIN_CATCH(h)
}

}  // extern "C"
