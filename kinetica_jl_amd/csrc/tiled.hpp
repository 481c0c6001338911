// "Library order" of a network for the tiled batched sweep (tiled_kernels.hip): a species order and a reaction order
// chosen by the library so that ONE pass over a state's rate constants needs random access to LDS only.
//
//   species   [ hubs | window 0 | window 1 | ... ]   hubs = the h most referenced species, resident in LDS (u and du)
//                                                      for the whole state; a window = a set of tail species whose u / du
//                                                      share the rest of the LDS while "their" records are processed
//   records   [ segment 0 | segment 1 | ... ]         record = a reaction and (if the network has it) its exact reverse;
//                                                      segment s holds the records whose tail species all lie in window s
//
// Such a partition exists whenever the graph "tail species that occur in the same record" falls apart into components
// smaller than a window: tail species are exactly the rarely referenced ones, so for sparse CRNs it does (the synthetic
// Zipf CRN at 50k species: largest component 282 species with 5 000 hubs, 17 with 10 000). A network whose tail does
// not decompose is reported as not tileable and keeps the hub / net-rate-scratch kernel (kernels.hip: sweep_big_kernel).
// A state that fits LDS entirely (N <= 10 000) is the special case h = N, one segment, species order = the caller's.
//
// Rate constants in library order: per segment, the first n2 records (its pairs, filled up to whole wavefronts with single
// reactions) have two slots each - forward / reverse rate constant, 0 where a reaction has no reverse - and the remaining
// records, all without a reverse, one: record i of a segment that starts at slot koff has its forward constant at
// koff + 2 i (i < n2) or koff + n2 + i (i >= n2). A network with (nearly) all its pairs has n2 = all records everywhere:
// k_lib[2 p] / k_lib[2 p + 1] for record p. The rate-table kernel writes this layout directly; Arrhenius parameters are
// stored per record for the sweep that forms its rate constants itself from the states' temperatures (SURVEY 8(d) M1').
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "network.hpp"

namespace kin {

constexpr int TILED_DUMMY = 64;        // per-lane dummy LDS entries (u = 1, du discarded) for unused record fields
constexpr int TILED_COPIES = 7;        // extra accumulator entries per split hub (copy 0 = the species' own entry)
constexpr int TILED_BATCH = 4;         // record rows of one batch of the kernel's load queue, at most (tiled_kernels.hip: NB)
constexpr int TILED_MIN_ROWS = 2 * TILED_BATCH;   // iteration rows of a segment, at least (the queue runs two batches ahead)
// The rows of a segment are padded (in iteration space only) to whole batches of the kernel that runs the layout: 4 rows
// for a windowed layout, 2 for a state that fits LDS whole (its staged-in set takes twice the registers). Records that
// touch hubs only go where they fill such a batch up (tiled.cpp).
constexpr int TILED_MAX_SEG = 48;              // segments a layout may have (their descriptors travel in the kernel arguments)
constexpr int TILED_EXP_TAB = 128;             // entries of the exp table the temperature form keeps in LDS (exp_tab.hpp)
constexpr int TILED_PIECE = 8192;                // species per piece of the staged layout conversion (64 kB of LDS: two workgroups per CU)
constexpr int TILED_LDS_ENTRIES = 10176;       // entries per LDS array (u, du): (2 x 10176 + 128) x 8 B = 160 kB exactly

struct TiledHost {
  bool ok = false;
  std::string why;                 // why the network is not tileable
  int32_t N = 0, R = 0, P = 0;     // species, reactions, records
  int32_t BS = 1024;               // workgroup size the layout was built for (a record row = BS records)
  int32_t h = 0;                   // hubs = library species [0, h), LDS entries [0, h)
  int32_t n_copy = 0;              // split-accumulator entries, LDS entries [h + 64, h + 64 + n_copy)
  int32_t wbase = 0;               // first LDS entry of the window region
  int32_t E = 0;                   // LDS entries per array
  int32_t T = 1;                   // segments (windows); T == 1 && win_cnt[0] == 0: no windows at all
  int32_t row_quantum = TILED_BATCH;   // iteration rows of every segment are a multiple of this (2 or 4, see above)
  bool identity = true;            // library species order == caller's
  std::vector<int32_t> species_of_lib, lib_of_species;   // N each
  // Layout conversion through LDS (tiled_kernels.hip: permute_staged_kernel; only when !identity). The caller's row is cut into
  // pieces of TILED_PIECE species; hubs and windows are each sorted by the caller's species index, so the species of a piece
  // that belong to one of them are a contiguous run of the library row. For the q-th element of that enumeration (piece after
  // piece, inside a piece hubs then window after window): stage_lib[q] = its library index, stage_off[q] = its offset inside
  // the piece. Piece c owns q in [c TILED_PIECE, min(N, (c + 1) TILED_PIECE)).
  std::vector<int32_t> stage_lib, stage_off;             // N each
  std::vector<int32_t> win_off, win_cnt;                 // T each: window s = library species [win_off, win_off + win_cnt)
  std::vector<int32_t> copy_src;   // n_copy: library index (= LDS entry) of the species behind each copy entry
  std::vector<uint32_t> rec;       // 2 words per record: four 14-bit LDS labels with fixed roles + flags in bits 56.. (tiled_kernels.hip)
  std::vector<int32_t> kf, kr;     // P each: reaction ids of a record's forward / reverse reaction (kr = -1: none)
  std::vector<int32_t> slot_of_reaction;   // R: position of reaction r's rate constant in a k_lib row
  std::vector<int32_t> rowtab;     // 2 per iteration row: first record of the row (-1: padding row), records in it
  std::vector<int32_t> seg_q;      // T + 1: iteration rows [seg_q[s], seg_q[s + 1]) belong to segment s (multiples of row_quantum)
  std::vector<int32_t> seginfo;    // 4 per segment: first record, records, iteration rows (>= TILED_MIN_ROWS), n2 - what the kernel reads
  std::vector<int32_t> seg_k;      // 2 per segment: koff = slot of its first record's forward constant (even), n2 = records with two slots
  std::vector<int32_t> kslot;      // P: slot of the record's forward constant (reverse: the next one); ~slot for a one-slot record
  int64_t KL = 0;                  // slots of a k_lib row (even)
  std::vector<int32_t> pad_slots;  // slots no record owns (a segment with an odd number of one-slot records ends in one): always 0
  bool has_singles = false;        // some segment has one-slot records (the kernel's SINGLES instantiation runs the layout)
  int64_t sched_slots = 0, sched_conflicts = 0;   // label slots placed / of those, on a bank an earlier lane of the 16-lane group uses
  int64_t k_len() const { return KL; }
};

// `bs`: workgroup size (256 / 512 / 1024). `h_force` > 0 fixes the hub count (tests).
TiledHost build_tiled(const NetworkHost& H, int bs, int h_force = 0);

}  // namespace kin
