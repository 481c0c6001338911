// Device view and launchers of the tiled batched sweep (tiled_kernels.hip; layout: tiled.hpp).
#pragma once
#include "common.hpp"
#include "tiled.hpp"

namespace kin {

struct TiledView {   // passed to the kernel by value
  int N, P, h, n_copy, wbase, E, T, win_cnt_max;
  int64_t KL;              // rate-constant slots of a k_lib row
  int has_singles;         // some segment ends in one-slot records (tiled.hpp)
  const uint2* rec;
  const int32_t* copy_src;
  // per segment, inside the kernel arguments (read with scalar loads from the kernarg segment, never through the
  // vector memory path): first record, records, iteration rows (>= TILED_GROUP, a multiple of it), two-slot records;
  // slot of the segment's first record; the window
  int4 seginfo[TILED_MAX_SEG];
  int32_t segk[TILED_MAX_SEG];
  int32_t win_off[TILED_MAX_SEG], win_cnt[TILED_MAX_SEG];
  const double4* par;      // Arrhenius parameters per record (TMODE and the library-order rate table)
  int has_kmax;
  double inv_kmax;
};

// Exactly one of k_lib (B x KL rate constants in library order) and Tb (B temperatures, device) is non-null.
// u, du: B x N in library species order. n_cu: compute units of the device the stream belongs to.
void launch_tiled_sweep(const TiledView& v, int bs, int n_cu, int64_t B, const double* u, const double* k_lib, const double* Tb,
                        double* du, hipStream_t s);
// dst[b][j] = map[j] >= 0 ? src[b][map[j]] : 0 for b < B (rows of n_dst / n_src doubles): the layout conversions
void launch_gather_rows(int64_t n_dst, int64_t n_src, int64_t B, const int32_t* map, const double* src, double* dst, hipStream_t s);
// The species permutation through LDS (tiled.hpp: stage_lib / stage_off): every global access is part of a coalesced stream -
// the caller's row piece by piece, the library row in runs of one hub / window range per piece. to_lib: dst (library order) from
// src (caller's order); else the reverse. B x N doubles each way.
void launch_permute_staged(int64_t N, int64_t B, bool to_lib, const int32_t* stage_lib, const int32_t* stage_off, const double* src,
                           double* dst, hipStream_t s);
// k_lib[b][2 p .. 2 p + 1] = k[b][kf[p]], k[b][kr[p]] (0 without a reverse): layouts whose records all have two slots; src rows of R
// doubles, R even and src 16-byte aligned
void launch_rates_to_lib_pairs(int P, int64_t R, int64_t B, const int32_t* kf, const int32_t* kr, const double* src, double* dst,
                               hipStream_t s);
void launch_tiled_params(int P, const int32_t* kf, const int32_t* kr, const double* Ea, const double* A, int has_kmax,
                         double t_mult, void* par, hipStream_t s);
struct TiledPadSlots { int n; int32_t slot[TILED_MAX_SEG]; };   // slots of a k_lib row that no record owns
void launch_rate_table_lib(int P, int64_t KL, int64_t n_stops, const void* par, const int32_t* kslot, const TiledPadSlots& pads,
                           int has_kmax, double k_max, const double* T, double* table, hipStream_t s);

}  // namespace kin
