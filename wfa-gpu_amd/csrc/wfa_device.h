// Shared device/host definitions for the gfx950 WFA kernels.
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//   * ASCII input:   the reference's padded buffer + 48-byte sequence_pair_t
//                    records (utils/sequences.h:28-36 of the reference).
//   * Packed input:  2 bits/base, code=(c&6)>>1, 16 bases per 32-bit word,
//                    base i of a sequence in bits [2*(i%16)+1 : 2*(i%16)]
//                    (little-endian inside the word so one v_alignbit_b32
//                    yields the 16 bases starting at any position).
//   * Backtrace arena: one origin byte per wavefront cell, bump-allocated in
//                    16-byte units.  Two layouts, told apart by the first word
//                    of a pair's block (bt_final_row[pair]):
//                    - TILES (the one-wave exact tier, round 6): 64-byte header
//                      {WFA_ROW_NONE, wlo, tile columns, budget}, then 64-byte
//                      tiles of 4 scores x 16 diagonals at a fixed pitch: tile
//                      (s >> 2, (k - wlo) >> 4), byte (s & 3) * 16 + ((k - wlo) & 15);
//                    - ROWS (every other tier): a row table [score] = {unit of
//                      the row, lo} (8 bytes per score) and one row of origin
//                      bytes per score, byte j = diagonal lo + j.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

// Mirror of the reference ABI struct sequence_pair_t (utils/sequences.h:28-36).
struct WfaSeqPair {
  size_t text_offset;
  size_t pattern_offset;
  size_t text_offset_packed;
  size_t pattern_offset_packed;
  unsigned int text_len;
  unsigned int pattern_len;
  bool has_N;
};
static_assert(sizeof(WfaSeqPair) == 48, "sequence_pair_t ABI");

// Per-pair status written by the align kernels.
enum : uint32_t {
  WFA_ST_PENDING = 0,    // not processed yet
  WFA_ST_DONE = 1,       // score (and backtrace rows) valid
  WFA_ST_BAND = 2,       // wavefront left the tier's diagonal capacity -> next tier
  WFA_ST_SCORE = 3,      // score went past the tier's score limit        -> next tier
  WFA_ST_NOMEM = 4,      // backtrace arena exhausted                     -> next sub-batch
  WFA_ST_ALPHABET = 5,   // pair holds bytes outside ACGT                 -> byte-compare tier
};

// Origin byte per wavefront cell (what the backtrace needs, 4 bits used):
//   bits 3:2  source of M : 0 none (cell not valid), 3 mismatch, 2 deletion, 1 insertion  (WFA2's priority on equal
//             offsets, wavefront_backtrace.c:48-59: the lean kernel path gets it from ONE signed max over
//             offset << 16 | these bits)
//   bit  1    D came from D (gap extension) rather than from M (gap open)
//   bit  0    I came from I (gap extension) rather than from M (gap open)
enum : uint32_t { BT_M_NONE = 0, BT_M_X = 12, BT_M_D = 8, BT_M_I = 4, BT_M_MASK = 12, BT_D_EXT = 2, BT_I_EXT = 1 };

// Operations of a reversed op list (what the backward walk leaves for the replay): the kind, and whether a run of matches
// follows the operation (re-derived by the replay as a longest common prefix).
enum : uint8_t { OP_X = 1, OP_I = 2, OP_D = 3, OP_EXT_AFTER = 0x10 };
// cigar_off of an alignment between walk and replay: byte offset of its op list -- in the op scratch, or (this bit set) in the
// backtrace ARENA, where the one-wave wavefront kernels leave it when they walk their own alignments (WfaAlignParams::walk_in_kernel)
#define WFA_OPS_IN_ARENA (1ull << 63)

#define WFA_ROW_NONE 0xFFFFFFFFu
// The lean cells of the 16-bit LDS tiers (0, 1, 2, 4) never switch a lane off: the lanes of a row's last 64-diagonal
// chunk that lie beyond its upper limit store NULL offsets into that many cells behind the row (every ring row carries
// WFA_RING_ROW_PAD extra cells for it) and origin bytes behind the row's bytes in the arena (every arena chunk keeps
// WFA_ARENA_ROW_SLACK 16-byte units unallocated at its end; the arena itself ends with 8 spare units).
#define WFA_RING_ROW_PAD 64
#define WFA_ARENA_ROW_SLACK 4u

// Backtrace arena, 16-byte units.  Per alignment: a block of tiles, or a row table [score] = {unit of the row, lo} (8 bytes
// per score up to the score budget) followed, wherever the block's bump allocator puts them, by one
// row of origin bytes per score (byte j = diagonal lo + j): the two layouts at the top of this file.
struct WfaAlignParams {
  const uint32_t* packed;        // packed sequences (word base); RAW kernels: the ASCII buffer
  const char* ascii;             // non-null (packed kernels only): the sequences are still ASCII -- pack them while they are staged: the words go
                                 // to LDS and to `packed` (for the backtrace kernels); a pair with a byte outside ACGT leaves with status
                                 // ALPHABET (counted in *n_raw) and is aligned by the byte-compare class afterwards
  unsigned long long* n_raw;
  const WfaSeqPair* meta;
  const uint32_t* work;          // pair indices to process (NULL: identity)
  uint32_t n_work;
  const unsigned long long* n_work_dev;   // optional: the list's real length lives on the device (a list compacted by the launch
                                          // before, not read back by the host yet); n_work is then an upper bound
  int only_pending;              // 1: skip pairs whose status is not PENDING (an unfiltered list: pairs flagged for the byte-compare class)
  unsigned int* work_counter;    // dynamic work distribution: 8 counters, 64 bytes apart (zeroed before launch)
  uint32_t work_shards;          // 1 or 8 of them in use
  int x, oe, e;                  // penalties: mismatch, open+extend, extend
  int dm, de;                    // ring depths: max(x,oe)+1 rows of M, e+1 rows of I and D
  int book_mask;                 // row-book entries - 1 (power of two >= max(dm, 64))
  int rs;                        // row stride (elements), even: widest diagonal window + 3
  int max_score;                 // give up (WFA_ST_SCORE) beyond this score
  const int32_t* budget;         // optional per-pair score budget (auto-tuned), capped by max_score
  int budget_q, budget_slack, budget_margin;   // tier 5 only, budget == NULL, budget_q > 0: the per-pair budget by k_budget's rule (wfa_host.hip), computed in the
                                 // kernel: (q * longer length * margin / 100) / 1024 + slack -- no launch to fill the array in front of the call's only wavefront launch
  int band_width;                // > 0: adaptive band, diagonals kept per wavefront (banded kernels)
  int band_period;               //      re-centre the band every this many scores
  int seq_words_cap;             // LDS words reserved per packed sequence
  int no_lean;                   // diagnostics (WFAGPU_NO_LEAN): keep every score on the careful path
  int32_t* score;                // [pair] out
  uint32_t* status;              // [pair] out
  uint32_t* cells;               // [pair] out, optional: number of wavefront cells computed
  unsigned long long* launch_cells;  // optional: += cells computed by this launch (one atomic per workgroup)
  // tier 5 only.  wave_parts: four u64 per wavefront of the launch {cells, pairs of its share of the list that did not leave DONE, pairs
  // it appended to fail_list, pairs it flagged ALPHABET}, plain stores -- to device memory, or straight to pinned host memory --
  // (instead of the atomic on launch_cells: thousands of wavefronts ending together on one counter line).  fail_list / fail_count: the
  // kernel appends the pairs it leaves with status BAND or SCORE itself (the chain's only launch, score-only: no kernel behind it)
  unsigned long long* wave_parts;
  uint32_t* fail_list;
  unsigned long long* fail_count;
  int arena_top_known;           // tier 5 with CIGARs: 1 = the bump pointer is arena_top0_value when the launch starts (the host knows: a pass's first
  unsigned long long arena_top0_value;   // launch) -- the launch moves it past its slots itself (workgroup 0) instead of a one-thread kernel behind it
  // backtrace (CIGAR mode)
  uint8_t* arena;                // base of the arena
  unsigned long long arena_units;        // capacity in 16-byte units
  unsigned long long* arena_top;         // bump pointer (units)
  uint32_t chunk_units;          // refill granularity
  uint32_t* bt_final_row;        // [pair] out: unit offset of the pair's block of tiles / row table
  // (round 5's option -- the one-wave tiers walking a finished alignment back themselves, a measured loss -- is gone: always 0)
  int walk_in_kernel;
  unsigned long long* cigar_off; // [pair] out (walk_in_kernel): WFA_OPS_IN_ARENA | byte offset of the op list in the arena
  uint32_t* cigar_len;           // [pair] out (walk_in_kernel): number of operations
  // global-memory ring (only the GLOBAL_RING instantiation)
  void* gring;                   // per-block slices of gring_stride bytes
  int ring16;                    //   16-bit offsets in it (sequences <= 32766 bases), else 32-bit
  unsigned long long gring_stride;
  // diagnostics (tuning.timed_barriers): barrier arrival / release clocks of workgroup 0, 3 x u64 per score and wave
  unsigned long long* dbg_times;
  uint32_t dbg_cap;              // records (scores) the buffer holds
};

struct WfaTraceParams {
  int raw;                       // 1: sequences are the ASCII buffer (byte compare), 0: 2-bit packed
  int seq_lds_stride;            // > 0: stage each lane's pair in LDS, this many (odd) words per lane
  int emit_pairs;                //      ... of the first emit_pairs lanes of a wavefront (the others idle: LDS per wavefront buys occupancy)
  const uint32_t* packed;
  const WfaSeqPair* meta;
  const uint32_t* work;          // pair indices (NULL: identity)
  uint32_t n_work;
  int x, oe, e;
  const int32_t* score;
  int32_t* score_fix;            // banded passes: scores are replaced by the cost of the emitted CIGAR where they differ
  const uint32_t* status;
  const uint8_t* arena;
  unsigned long long arena_bytes;  // size of the arena (the wave-per-alignment kernel range-checks row-table entries of scores without a wavefront)
  int wave_kernel;               // 1: one wavefront per alignment (long alignments), 0: one lane per alignment
  int group;                     // wave kernel: alignments per wavefront (1, 2, 4 or 8: 64/group lanes each, own LDS share)
  int lane_fused;                // lane-per-alignment path: walk + replay in ONE kernel, op lists of ops_lds_bytes per lane in LDS (short alignments)
  int walk_grid_cap;             // workgroups of wfa_walk_kernel (it strides over the list); 0: one per 256 list entries
  int skip_walk;                 // lane path: every finished pair of the list was walked by its wavefront kernel: no wfa_walk_kernel
  int walk_only;                 // wave kernel: only the walk (op list -> slot w of ops_slot bytes in the global scratch); wfa_emit_kernel replays
  uint32_t ops_slot;
  int seq_words_cap;             // wave kernel: LDS words reserved per sequence
  int ops_lds_bytes;             // wave kernel: LDS bytes for the op list (longer lists go through the global scratch)
  int text_lds_bytes;            // wave kernel: LDS bytes for the CIGAR text of one alignment (longer texts: second replay)
  const uint32_t* bt_final_row;
  // scratch for the reversed op list, bump allocated (bytes)
  uint8_t* ops;
  unsigned long long ops_cap;
  unsigned long long* ops_top;
  // output text arena
  char* text;
  unsigned long long text_cap;
  unsigned long long* text_top;
  // lane-per-alignment emit, single replay: texts are first written at upper-bound offsets of this scratch and then
  // compacted into `text` (NULL: two replays, the first one only to size the text)
  char* text_scratch;
  unsigned long long text_scratch_cap;
  unsigned long long* scratch_top;
  int min_op_cost;               // min(x, e): an alignment of score s has at most s / min_op_cost operations
  int item_chars;                // widest RLE item (digits of the longest sequence + 1): sizes the upper-bound text slots
  unsigned long long* cigar_off; // [pair] out: byte offset of the CIGAR in `text`
  uint32_t* cigar_len;           // [pair] out: strlen; 0xFFFFFFFF if it did not fit
};

// Host-side launchers (one per translation unit).
// Timed launches.  hipEventRecord puts a barrier packet of its own into the queue, and a kernel behind one starts ~5.5 us after the
// kernel in front of it has finished (back to back: 0.2 us): nine records per call were 35 of the 250 us of a BASELINE configs[1]
// step.  hipExtLaunchKernelGGL lets the kernel's own dispatch packet carry the time stamps: ev0 becomes the start of the kernel,
// ev1 its end (either may be null; hipEventElapsedTime pairs them freely, also with recorded events: scratch/ext_event_probe.hip).
template <typename K, typename... Args>
inline void wfa_launch_timed(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, Args... args) {
  if (ev0 || ev1) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, ev0, ev1, 0, args...);
  else hipLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, args...);
}

// (ev0 / ev1 of every launcher: events that receive the start of the first and the end of the last kernel it launches)
void wfa_launch_pack(const char* d_ascii, const WfaSeqPair* d_meta, uint32_t n_pairs,
                     uint32_t* d_packed, uint8_t* d_flags, hipStream_t stream, uint32_t max_seq_len = 0,
                     hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                     uint32_t* d_status = nullptr, unsigned long long* d_n_raw = nullptr);      // (d_status: flagged pairs get WFA_ST_ALPHABET and are counted in *d_n_raw)
// tier: 0 -> 1 wave/alignment (LDS ring), 1 -> 4 waves (LDS), 2 -> 16 waves (LDS), 4 -> 16 waves, M and I rings in LDS + D ring in HBM,
//       3 -> 16 waves, ring in HBM (int16 or int32 offsets, WfaAlignParams::ring16).  Returns the dynamic LDS bytes used.
size_t wfa_align_lds_bytes(const WfaAlignParams& p, int tier);
// wpe: waves per SIMD the one-wave exact kernels (tier 0, packed class) are compiled for: 8, 7, 6 or 4 (others: 8).
void wfa_launch_align(const WfaAlignParams& p, int tier, bool with_bt, bool raw, int grid, hipStream_t stream, int wpe = 8,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int wfa_align_max_blocks_per_cu(int tier, bool with_bt, bool raw, bool banded, size_t lds_bytes, int wpe = 8);
bool wfa_launch_trace(const WfaTraceParams& p, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);     // false: nothing to launch
// Tier 5 (short_kernel.hip): 64 / lanes alignments per wavefront (lanes = 16 or 32 diagonals each), rings in registers; with_bt: one row
// of `lanes` origin bytes per score for the backtrace.
bool wfa_short_supported(int x, int oe, int e);
size_t wfa_short_lds_bytes(const WfaAlignParams& p, int lanes, bool with_bt);
void wfa_launch_short(const WfaAlignParams& p, int lanes, bool with_bt, int grid, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int wfa_short_max_blocks_per_cu(const WfaAlignParams& p, int lanes, bool with_bt);     // (wavefronts a CU holds: registers and LDS)
unsigned long long wfa_short_bt_slot_units(int max_score, int lanes);     // (with_bt: arena units every work item of the launch owns)
// Tier 5's instantiations are picked from tables [x - 1][o + e - 1]; one translation unit (one code object) per gap extension e:
// short_kernel.hip holds e = 1, short_kernel_e2/3/4.hip the others (short_kernel_impl.h is the kernel).
struct WfaShortEntry { void (*launch)(const WfaAlignParams&, int, hipStream_t, hipEvent_t, hipEvent_t); int (*occ)(size_t); };
const WfaShortEntry* wfa_short_entry_e2(int ascii, int bt, int l32, int idx);
const WfaShortEntry* wfa_short_entry_e3(int ascii, int bt, int l32, int idx);
const WfaShortEntry* wfa_short_entry_e4(int ascii, int bt, int l32, int idx);
// Code-object priming: one empty launch per kernel translation unit (see the definitions).
void wfa_prime_pack(hipStream_t stream);
void wfa_prime_align(hipStream_t stream);
void wfa_prime_short(hipStream_t stream);
void wfa_prime_trace(hipStream_t stream);
