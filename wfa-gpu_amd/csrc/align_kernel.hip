// Gap-affine wavefront alignment (forward pass) for gfx950.
//
// Replaces the reference's distance_kernel / alignment_kernel and their adaptive-band variants
// (lib/kernels/sequence_distance_kernel.cu:175-425, sequence_alignment_kernel.cu:355-688,
//  sequence_*_kernel_aband.cu), its extend device function
// (lib/kernels/common_alignment_kernels.cuh:29-111) and their launch layer
// (lib/sequence_alignment.cu:211-470).  The arithmetic follows WFA2-lib, the ground truth the
// reference checks itself against (paths under external/WFA/wavefront):
//   recurrences + out-of-range nulling  wavefront_compute_affine.c:45-87
//   per-component end trimming          wavefront_compute.c:570-603
//   limits of the next wavefront        wavefront_compute.c:41-71
//   termination                         wavefront_extend.c:47-67
//   tie-breaks recorded for backtrace   wavefront_backtrace.c:48-59,366-376
// so that score AND CIGAR are identical to WFA2's.
//
// Design (MI355X):
//   * A persistent workgroup of NW wavefronts (NW = 1, 4 or 16) owns one alignment at a time and
//     claims the next one from 8 sharded counters.  The diagonals of the current score are striped
//     over the NW*64 lanes.
//   * The wavefront ring -- max(x,o+e)+1 rows of M, e+1 rows of I and of D, 16-bit offsets -- and
//     both 2-bit packed sequences live in LDS.  Only the diagonals an alignment within the score
//     budget can visit are kept (see "window" below), which halves the ring against the
//     reference's |k| <= max_error sizing.  Wider wavefronts: the hybrid tier keeps the M and I rings in LDS
//     and the D ring in HBM; the last resort keeps the whole ring (16- or 32-bit) in HBM/L2.
//   * Ring invariant: a row holds NULL everywhere outside the limits it was last written with (set up
//     once per alignment, kept by clearing what a slot's previous occupant had beyond the new limits),
//     so the five reads per cell need no range predicate.  No lane is switched off: in the careful loop lanes past
//     the end of a row recompute its last cell, in the lean loops they store NULL into the padding behind the row.
//   * Two score loops.  The CAREFUL one is WFA2 to the letter (limits from the trimmed limits of the input
//     rows, "no wavefront" scores, trimming of values past a sequence end, termination read per score).  The
//     LEAN one runs until an M cell first touches a sequence end -- before that no value can run past an
//     end, nothing is trimmed and no cell can be the last one.  Its cells (hot_cells) address every row as one
//     lane base + immediates (bases formed once per score), break ties with one signed max over
//     (offset << 16 | origin bits), give NULL cells a run length of 0 through v_med3 instead of selecting, extend
//     without exec mask (first 16-base step straight-line) and take the run limit from an LDS row; its per-score part
//     has closed-form limits (e == 1; for e > 1 a superset of WFA2's limits carried by two counters), in-place state, one loop
//     exit, and touches the row book only when it leaves.
//     Instruction issue (vector pipe 89 % busy on the headline workload, scalar pipe 71 %) is what this
//     kernel is bound by; the lean loop exists to issue fewer instructions per cell and per score.
//   * NW == 1: no barrier anywhere in the score loop (LDS operations of one wavefront execute in
//     order) and the per-score bookkeeping lives in three VGPRs indexed by lane (v_readlane), not in
//     memory; the exact one-wave kernels exist compiled for 8, 7, 6 and 4 waves per SIMD (WPE) and the host
//     picks the one that matches the rings LDS lets a CU hold.  NW > 1: one barrier per score.
//   * Score loops: the careful one; the closed-form lean loop (exact LDS tiers, e == 1); the general lean loop
//     (LDS tiers, any gap extension, and the adaptive band); one lean loop for the tiers whose ring lives in
//     HBM.  They share the cells and refill_arena (the one place arena space is claimed); the banded search has one loop
//     of its own over the same cells.
//   * Adaptive band (BANDED): a row holds band_width diagonals, stored relative to its own lower limit
//     between two NULL guard zones -- per-row scalar offsets on the row addresses, the same cells and loops
//     as the exact search; the band moves by at most two diagonals per score.
//   * extend(): two 32-bit LDS words per sequence, v_alignbit_b32 to the base position, XOR,
//     count-trailing-zeros: 16 bases per iteration (4 bytes in the byte-compare instantiation).
//   * For CIGARs each cell emits ONE origin byte into a bump-allocated arena.  The one-wave exact tier (round 6): into a block
//     of 64-byte TILES of 4 scores x 16 diagonals that the alignment claims once (TILED, below: the backward walk touches half
//     the cache lines, a score sizes / claims / records nothing); the other tiers: one row per score (64 consecutive bytes per
//     wavefront store) behind a per-alignment row table (8 bytes per score).  Nothing to memset between alignments.
//
// Files: this one holds the kernel's skeleton -- LDS layout, work claiming, per-alignment set-up (window, staging, ring reset,
// score 0), the ring state shared by all score loops, the epilogue -- and the launchers.  The score loops and the cells are
// pieces of the kernel's body kept in files of their own (textually included where they run; each says what it reads and writes):
//   align/cells_hot.inc       the lean cells of the 16-bit LDS tiers: the hot loop of the library
//   align/cells_generic.inc   the cells as WFA2 has them (careful loop, HBM-ring tiers, banded search after a jump)
//   align/loop_lean_e1.inc    lean score loop, gap extension 1, closed-form limits
//   align/loop_lean_any.inc   lean score loop, gap extension > 1: superset limits from two counters (round 6)
//   align/loop_lean_hbm.inc   lean score loop of the tiers whose ring lives in HBM
//   align/loop_careful.inc    the careful score step (WFA2 to the letter)
//   align/band_window.inc, align/loop_banded.inc   the adaptive band: the reference's window rule, the banded search
#include <type_traits>

#include "wfa_device.h"
#include "pack_device.h"

namespace {

// "No cell": any negative offset.  A NULL that is read as a predecessor is incremented once per score along the edges of the
// wavefront, so it must sit further below zero than an alignment has scores: 16-bit rows serve scores <= 30000, the 32-bit
// rows (HBM ring of the unbounded tier) get a quarter of the int range.
template <typename OffT> struct OffNull { static constexpr int value = -32768; };
template <> struct OffNull<int32_t> { static constexpr int value = INT_MIN / 4; };

template <typename OffT> __device__ __forceinline__ OffT off_store(int v);
template <> __device__ __forceinline__ int16_t off_store<int16_t>(int v) {
  // an I chain running past the end of the text only ever grows; saturate it so that it stays
  // "past the end" in 16 bits (host guarantees lengths <= 32766)
  return (int16_t)min(v, 32767);
}
template <> __device__ __forceinline__ int32_t off_store<int32_t>(int v) { return v; }

// LDS byte address of a pointer into the dynamic shared array (for hand-written DS instructions)
template <typename T> __device__ __forceinline__ uint32_t lds_addr(const T* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

template <int NW> __device__ __forceinline__ void block_sync() {
  if constexpr (NW == 1) {
    // single wavefront: DS/VMEM operations issue in program order, only the compiler has to be
    // stopped from reordering or caching across this point
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// value of thread 0 -> every thread of the block
template <int NW> __device__ __forceinline__ uint32_t block_bcast(uint32_t v, uint32_t* slot) {
  if constexpr (NW == 1) {
    return __builtin_amdgcn_readfirstlane(v);
  } else {
    if (threadIdx.x == 0) *slot = v;
    __syncthreads();
    const uint32_t r = *slot;
    __syncthreads();
    return r;
  }
}

// Per-score bookkeeping of the ring, indexed by (score & book_mask):
//   A    = limits of the row as computed, lo in the low and hi in the high 16 bits (lo > hi: no wavefront)
//   I, D = limits of the I and D components (the computed ones, or the trimmed ones when a value ran
//          past a sequence end; lo = 1, hi = -1 when the component does not exist)
// NW == 1 keeps each in one VGPR (lane = slot): reading is one v_readlane, no LDS round trip, and
// everything derived from it stays on the scalar unit.
// "No wavefront / no component": lo far above hi, so that min(lo, lo' - 1) / max(hi, hi' + 1) over a mix of existing and
// missing rows ignore the missing ones by themselves (WFA2's null wavefront carries lo = 1, hi = -1, whose traces in the
// limits of the next wavefront only ever add cells that are not valid).
constexpr int ROW_NONE_LO = 16000, ROW_NONE_HI = -16000;
constexpr int ROW_NONE_A = (int)(((unsigned)ROW_NONE_LO & 0xFFFFu) | ((unsigned)ROW_NONE_HI << 16));
__device__ __forceinline__ int pack_range(int lo, int hi) { return (lo & 0xFFFF) | (int)((unsigned)hi << 16); }
__device__ __forceinline__ int range_lo(int a) { return (int)(int16_t)(a & 0xFFFF); }
__device__ __forceinline__ int range_hi(int a) { return a >> 16; }

template <int NW> struct RowBook {
  int a, i, d;          // NW == 1
  int *A, *I, *D;       // NW > 1 (LDS)
  __device__ __forceinline__ int get_a(int slot) const { if constexpr (NW == 1) return __builtin_amdgcn_readlane(a, slot); else return A[slot]; }
  __device__ __forceinline__ int get_i(int slot) const { if constexpr (NW == 1) return __builtin_amdgcn_readlane(i, slot); else return I[slot]; }
  __device__ __forceinline__ int get_d(int slot) const { if constexpr (NW == 1) return __builtin_amdgcn_readlane(d, slot); else return D[slot]; }
  __device__ __forceinline__ void reset() { a = i = d = ROW_NONE_A; }
  // Lean path: every score it records has I and D limits equal to the A limits, and so have the dm scores before it
  // (they were regular); older entries are never read again.  So only A is recorded per score and I, D are set from A
  // wholesale when the path is left.
  __device__ __forceinline__ void set_a(int slot, int va) {
    if constexpr (NW == 1) { a = ((int)(threadIdx.x & 63) == slot) ? va : a; } else { A[slot] = va; }
  }
  __device__ __forceinline__ void copy_a_to_id(int tid, int nt, int mask) {
    if constexpr (NW == 1) { i = a; d = a; }
    else { __syncthreads(); for (int j = tid; j <= mask; j += nt) { I[j] = A[j]; D[j] = A[j]; } __syncthreads(); }
  }
  // every thread calls set with the same values (NW > 1: same-value stores, each thread reads back its own)
  __device__ __forceinline__ void set(int slot, int va, int vi, int vd) {
    if constexpr (NW == 1) {
      const bool mine = (int)(threadIdx.x & 63) == slot;
      a = mine ? va : a; i = mine ? vi : i; d = mine ? vd : d;
    } else {
      A[slot] = va; I[slot] = vi; D[slot] = vd;
    }
  }
};

// Kernel arguments that are only needed between alignments (work list, result arrays, arena
// bookkeeping) are re-read from the kernarg segment where they are used instead of being held in
// SGPRs across the score loop: the loop needs every scalar register it can get (spilled SGPRs come
// back through v_readlane, i.e. through the vector unit this kernel saturates).
typedef const WfaAlignParams __attribute__((address_space(4)))* ColdParams;
__device__ __forceinline__ ColdParams cold_params() {
  unsigned long long v = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(v));
  return (ColdParams)v;
}

// HYBRID: the M and I rings in LDS, the D ring (2 of the 9 rows at the default penalties) in global memory -- for
// wavefronts whose whole ring misses the 160 KiB of a CU by a little (30 kbp at 10 % error: 9 rows x 18 KB).  The
// all-global ring moves 16 bytes per cell through L2; this one 4.
// WPE: waves per SIMD the one-wave exact kernels are compiled for.  Their residency is set by LDS (one ring per wave):
// where that leaves 7 or fewer waves per SIMD, the register diet of 8 (64 VGPRs, 78 SGPRs: 123 scalar spills, each a
// v_readlane/v_writelane on the pipe the kernel saturates, and scratch) buys nothing; the host picks the instantiation
// that matches the rings a CU holds (plan_tier).
// TIMED (diagnostics, tuning.timed_barriers): workgroup 0 records, for every score of the closed-form lean loop, the s_memtime
// at which each of its waves reaches the per-score barrier and the one at which it leaves it (profiles/r04/barrier_skew.md).
template <int NW, bool BT, typename OffT, bool GLOBAL_RING, bool RAW, bool BANDED, bool HYBRID = false, int WPE = 8, bool TIMED = false>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((NW == 1 && !BANDED) ? WPE : 1, 8)))
wfa_align_kernel(const WfaAlignParams p) {
  static_assert(!HYBRID || (!GLOBAL_RING && !BANDED), "the hybrid ring is an exact LDS tier");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // (which 64-diagonal stripe of a row this thread's wave takes in the lean cells: wave order.  Reversing it for the four-wave
  // tier -- so that wave 0, which also keeps the row table and is the last at the barrier in 48 % of the scores,
  // profiles/r04/barrier_skew.md, gets the fewest chunks -- measured 13.0 -> 13.3 ms on BASELINE configs[3]: not kept.)
  const int stid = tid;
  const int dm = p.dm, de = p.de, rs = p.rs;
  constexpr int ROW_PAD = (!GLOBAL_RING && sizeof(OffT) == 2) ? WFA_RING_ROW_PAD : 0;     // (see wfa_device.h)
  // the exact tiers with 16-bit offsets in LDS (0, 1, 2 and the hybrid ring, whose D rows live in global memory): their
  // lean loops have a form of their own; HM_ROW: the one-wave tier also keeps min(plen + k, tlen) per diagonal in LDS (the
  // multi-wave tiers are LDS-bound: 16k x 10 kbp, four waves: 6 rings per CU without the row, 5 with it, 16.4 vs 17.7 ms)
  // BANDED (adaptive band, a heuristic): every score keeps band_width diagonals, chosen by the reference's rule (band_window
  // below), and a ring row holds just those: the row of a score is stored RELATIVE to its own lower limit (column GZ = its
  // diagonal lo), so the "diagonal 0" address of a row is its slot base plus a per-row scalar offset -- the cells (lean and
  // generic) are shared with the exact search.  While the rows a cell reads lie within GZ - 1 diagonals of its own row, every
  // read lands in a row's cells or in its guard zones (GZ columns on either side, NULL) and needs no range test; after a
  // re-centring jump the generic cells range-check every read.
  constexpr bool HOT = !GLOBAL_RING && sizeof(OffT) == 2;
  constexpr bool HM_ROW = HOT && !HYBRID && NW == 1 && !BANDED;
  // TILED (round 6): the origin bytes of the one-wave exact tier go into a block of TILES that the alignment owns as a whole --
  // 64-byte tiles of 4 scores x 16 diagonals, tile (s >> 2, (k - wlo) >> 4) at a fixed pitch, byte (s & 3) * 16 + ((k - wlo) & 15) -- instead
  // of one bump-allocated row per score behind a row table (wfa_device.h).  The backward walk of a 1 kbp alignment then touches ~35
  // cache lines instead of 74 (55 origin bytes on 55 lines + 19 lines of row table: wfa_walk_kernel is bound by random 64-byte
  // accesses), and a score of the lean loops neither sizes, claims nor records a row: no refill test, no row-table entry.
  constexpr bool TILED = BT && HOT && !HYBRID && NW == 1 && !BANDED;
  const int GZ = BANDED ? 4 * dm + 2 : 0;
  const int x = p.x, oe = p.oe, e = p.e;

  // ---- carve LDS -----------------------------------------------------------------------------
  unsigned char* sp = smem;
  OffT* Mr;
  if constexpr (GLOBAL_RING) {
    Mr = reinterpret_cast<OffT*>(static_cast<char*>(p.gring) + (size_t)blockIdx.x * p.gring_stride);
  } else {
    Mr = reinterpret_cast<OffT*>(sp);
    // (HM_ROW: one more row, min(plen + k, tlen) per diagonal k -- how far a run on k can go)
    sp += (((size_t)(dm + (HYBRID ? 1 : 2) * de + (HM_ROW ? 1 : 0)) * rs * sizeof(OffT)) + 15) & ~(size_t)15;
  }
  OffT* const Dg = HYBRID ? reinterpret_cast<OffT*>(static_cast<char*>(p.gring) + (size_t)blockIdx.x * p.gring_stride) : nullptr;   // the D ring
  uint32_t* Pw = reinterpret_cast<uint32_t*>(sp);
  uint32_t* Tw = Pw + p.seq_words_cap;
  const int bkm = p.book_mask;                                // row book: 64 (or more) entries indexed by score & bkm
  int* red = reinterpret_cast<int*>(Tw + p.seq_words_cap);    // [3][8] per-score reduction slots (NW > 1)
  uint32_t* bslot = reinterpret_cast<uint32_t*>(red + 24);    // [2] broadcast slots
  // "A cell of this score touched a sequence end" (multi-wave lean loops): one LDS word PER SCORE PARITY, set by the waves that see it,
  // read by every wave behind the score's barrier.  With a single sticky word a wave that is already computing score s + 1 could set it
  // before a slower wave had done its read for score s -- the waves would then leave the loop at different scores with their barrier
  // counts still matching (silent corruption; never observed, not excluded: ADVICE r5).  A wave cannot be two scores ahead of another
  // one (the barrier of s + 1 stands between), so two words do.  They are words 7 of the reduction slots 0 and 1, which the careful
  // loop never uses (it works on words 0..6 and re-initialises all 24 when a lean loop hands over); the banded search, which never
  // runs the careful loop but scans re-centrings through red[0 .. NW-1], takes words 22 and 23.
  auto touch_flag = [&](const int score) -> uint32_t* { return reinterpret_cast<uint32_t*>(red + (BANDED ? 22 + (score & 1) : 7 + 8 * (score & 1))); };
  // (the banded search keeps the book in registers in every tier: all waves of a workgroup derive the same windows, each keeps
  // its own copy -- no LDS round trip, no barrier for the book: loop_banded.inc)
  constexpr int BOOK_NW = BANDED ? 1 : NW;
  RowBook<BOOK_NW> book;
  book.reset();
  if constexpr (BOOK_NW == 1) { book.A = book.I = book.D = nullptr; }
  else { book.A = reinterpret_cast<int*>(bslot + 2); book.I = book.A + (bkm + 1); book.D = book.I + (bkm + 1); }

  uint32_t chunk_cur = 0, chunk_left = 0;   // arena units owned by this block
  uint32_t dbg_i = 0;                       // (TIMED: barrier records written by this wave so far)
  unsigned long long blk_cells = 0;         // cells computed by this workgroup (reported once, at the end)

  // Work distribution: the list is cut into contiguous shards, each with its own counter on its
  // own cache line; a block starts on shard blockIdx % 8 and moves on when a shard is empty.  One
  // counter word serves only ~88 claims per microsecond, which capped short-read batches
  // (BASELINE configs[1]) at 1.2 ms per 100k pairs.
  uint32_t shard = blockIdx.x & (cold_params()->work_shards - 1u), shards_left = cold_params()->work_shards;   // 1 or 8
  for (;;) {
    uint32_t w = 0xFFFFFFFFu;
    {
      ColdParams cp = cold_params();
      const uint32_t nsh = cp->work_shards;
      uint32_t n_work = cp->n_work;
      { const unsigned long long* nd = cp->n_work_dev; if (nd) n_work = min(n_work, (uint32_t)*nd); }      // (list length still on the device)
      while (shards_left) {
        const uint32_t lo_w = (uint32_t)(((unsigned long long)n_work * shard) / nsh);
        const uint32_t hi_w = (uint32_t)(((unsigned long long)n_work * (shard + 1)) / nsh);
        // (a claim is a returning atomic on one word, ~88 per microsecond: an EMPTY shard -- the speculative re-run launch of a
        // chain mostly finds nothing at all -- is skipped without one: 8192 workgroups x 8 shards of atomics were 0.17 ms of an
        // empty launch.  Looking at the counter with a plain load before every claim was tried too: it costs the full launch
        // of BASELINE configs[2] 2.6 %, a round trip per pair.)
        const uint32_t size_w = hi_w - lo_w;
        if (size_w != 0u) {
          uint32_t c = 0;
          if (tid == 0) c = atomicAdd(cp->work_counter + shard * 16, 1u);
          c = block_bcast<NW>(c, bslot);
          if (c < size_w) { w = lo_w + c; break; }
        }
        shard = (shard + 1) & (nsh - 1u); --shards_left;
      }
    }
    if (w == 0xFFFFFFFFu) break;
    const uint32_t* work = cold_params()->work;
    // (every thread loads the same words; readfirstlane tells the compiler so)
    const uint32_t pair = __builtin_amdgcn_readfirstlane(work ? work[w] : w);
    if (cold_params()->only_pending && __builtin_amdgcn_readfirstlane(cold_params()->status[pair]) != WFA_ST_PENDING) continue;   // (uniform)
    const WfaSeqPair mp = cold_params()->meta[pair];
    const int plen = __builtin_amdgcn_readfirstlane((int)mp.pattern_len), tlen = __builtin_amdgcn_readfirstlane((int)mp.text_len);
    const int kend = tlen - plen;
    const int pwords = RAW ? ((plen + 3) >> 2) + 1 : ((plen + 15) >> 4) + 1;
    const int twords = RAW ? ((tlen + 3) >> 2) + 1 : ((tlen + 15) >> 4) + 1;

    typedef __attribute__((address_space(1))) uint8_t* GlobalBytes;
    uint32_t status = WFA_ST_DONE;
    int s = 0;
    uint32_t ncells = 1;
    bool done = false;
    uint32_t row_s = WFA_ROW_NONE, tab_base = WFA_ROW_NONE;
    uint2* tab = nullptr;                   // backtrace row table of this pair: [score] = {arena unit, lo}

    // Diagonal window that can hold an alignment of score <= max_score (exact, not a heuristic):
    // a path that visits diagonal k beyond both 0 and kend needs one gap out and one gap back,
    // i.e. costs at least 2o + (|k| + |k - kend|) e, so cells outside the window cannot lie on any
    // path the backtrace can choose while the score stays within the limit.
    // The budget may be per pair (host auto-tuning from a scored sample): smaller budget, narrower
    // window, and -- below -- a wavefront that shrinks again once the score passes half the budget.
    int budget = cold_params()->max_score;
    { const int32_t* pb = cold_params()->budget; if (pb) budget = min(budget, pb[pair]); }
    {
      // no optimal alignment costs more than min(P,T) mismatches plus one gap of |T - P| (that alignment
      // always exists); it bounds the backtrace row table
      const long long worst = (long long)x * min(plen, tlen) + (kend ? oe + (long long)e * (abs(kend) - 1) : 0);
      budget = (int)min((long long)budget, worst);
    }
    int wlo = -plen, whi = tlen;
    bool feasible = true;
    {
      if (budget < INT_MAX / 2) {
        const long long S = budget, o = oe - e;
        const int ak = kend < 0 ? -kend : kend;
        feasible = (ak ? o + (long long)ak * e : 0) <= S;
        const long long a_hi = S - 2 * o + (long long)kend * e, a_lo = S - 2 * o - (long long)kend * e;
        const int kmax0 = max(0, kend), kmin0 = min(0, kend);
        whi = a_hi >= 0 ? max(kmax0, (int)min((long long)tlen, a_hi / (2 * e))) : kmax0;
        wlo = a_lo >= 0 ? min(kmin0, -(int)min((long long)plen, a_lo / (2 * e))) : kmin0;
        whi = min(whi, tlen); wlo = max(wlo, -plen);
      }
    }
    const int kidx0 = BANDED ? 0 : dm - wlo;             // exact mode: row index of diagonal 0 (dm guard cells each side: the lean path
                                            // re-NULLs up to dm cells beyond a row's ends without looking anything up)

    if (!feasible) {
      status = WFA_ST_SCORE;
    } else if ((!BANDED && whi - wlo + 1 + 2 * dm + ROW_PAD > rs) || pwords > p.seq_words_cap || twords > p.seq_words_cap) {
      status = WFA_ST_BAND;
    } else {
      // ---- stage the sequences, reset the ring ---------------------------------------------
      const uint32_t* packed = cold_params()->packed;
      const uint32_t* __restrict__ gp = packed + ((RAW ? mp.pattern_offset : mp.pattern_offset_packed) >> 2);
      const uint32_t* __restrict__ gt = packed + ((RAW ? mp.text_offset : mp.text_offset_packed) >> 2);
      const char* const ascii = RAW ? nullptr : cold_params()->ascii;
      if (!RAW && ascii != nullptr) {
        // The batch is still ASCII (a resident batch that nobody packed): pack while staging -- 16 bytes in, one word to LDS and the
        // same word to the packed buffer, where the backtrace kernels will look for it.  (A pack kernel of its own reads 2 GB and
        // writes 0.5 GB per 1M x 1 kbp pairs before the first wavefront kernel can start: 0.63 ms; here the bytes arrive under the
        // arithmetic of the other wavefronts.)  Every launch does it for the pairs it stages: packing twice is harmless.
        const uint32_t* pa = plen ? reinterpret_cast<const uint32_t*>(ascii + mp.pattern_offset) : packed;      // (empty: fully masked loads, any safe address)
        const uint32_t* ta = tlen ? reinterpret_cast<const uint32_t*>(ascii + mp.text_offset) : packed;
        uint32_t* const wp = const_cast<uint32_t*>(gp); uint32_t* const wt = const_cast<uint32_t*>(gt);
        uint32_t bad = 0;
        for (int i = tid; i < pwords; i += NT) { const uint32_t w = wfa_pack::pack_word(wfa_pack::load_word(pa, (uint32_t)plen, (uint32_t)i), (uint32_t)plen, (uint32_t)i, bad); Pw[i] = w; wp[i] = w; }
        for (int i = tid; i < twords; i += NT) { const uint32_t w = wfa_pack::pack_word(wfa_pack::load_word(ta, (uint32_t)tlen, (uint32_t)i), (uint32_t)tlen, (uint32_t)i, bad); Tw[i] = w; wt[i] = w; }
        bool any_bad;
        if constexpr (NW == 1) any_bad = __builtin_amdgcn_ballot_w64(bad != 0u) != 0ull;
        else {
          // (through a word of the workgroup's own LDS: __syncthreads_or brings 256 bytes of STATIC LDS with it, and a ring that
          // fills the CU's LDS to the last bytes no longer fits -- found by the soak: the occupancy query of such a launch failed)
          if (tid == 0) bslot[1] = 0u;
          __syncthreads();
          if (bad != 0u) bslot[1] = 1u;
          __syncthreads();
          any_bad = __builtin_amdgcn_readfirstlane(bslot[1]) != 0u;      // (the same in every thread: a scalar)
        }
        if (any_bad) {
          // WFA2 compares raw bytes: this pair belongs to the byte-compare class, which runs after the packed one
          if (tid == 0) {
            ColdParams cp = cold_params();
            atomicAdd(cp->n_raw, 1ull);
            cp->score[pair] = -1;
            cp->status[pair] = WFA_ST_ALPHABET;
            if (cp->cells) cp->cells[pair] = 0;
          }
          block_sync<NW>();
          continue;
        }
      } else {
        for (int i = tid; i < pwords; i += NT) Pw[i] = gp[i];
        for (int i = tid; i < twords; i += NT) Tw[i] = gt[i];
      }
      {
        // Ring invariant: a row holds NULL everywhere outside the limits it was last written
        // with, so reads next to a row's ends need no predicate and no per-score guard fill.  It starts
        // here (rows of M, I, D are contiguous) and is kept by clearing, whenever a row is overwritten,
        // what the previous occupant of its slot had beyond the new limits.
        const int cells = (dm + (HYBRID ? 1 : 2) * de) * rs;     // rs is even
        if constexpr (sizeof(OffT) == 2) {
          uint32_t* w = reinterpret_cast<uint32_t*>(Mr);
          for (int i = tid; i < (cells >> 1); i += NT) w[i] = 0x80008000u;
          if constexpr (HYBRID) {
            uint32_t* wd = reinterpret_cast<uint32_t*>(Dg);
            for (int i = tid; i < ((de * rs) >> 1); i += NT) wd[i] = 0x80008000u;
          }
        } else {
          for (int i = tid; i < cells; i += NT) Mr[i] = (OffT)OffNull<OffT>::value;
        }
        if constexpr (HM_ROW) {
          OffT* hm = Mr + cells;        // [kidx0 + k] = min(plen + k, tlen)
          for (int i = tid; i < rs; i += NT) hm[i] = (OffT)max(min(plen + (i - kidx0), tlen), 0);
        }
      }
      if constexpr (BOOK_NW == 1) book.reset();
      else { for (int i = tid; i <= bkm; i += NT) { book.A[i] = book.I[i] = book.D[i] = ROW_NONE_A; } }
      if constexpr (NW > 1) { if (tid < 24) red[tid] = (tid & 7) == 6 ? 0 : (((tid & 7) & 1) ? INT_MIN : INT_MAX); }
      block_sync<NW>();

      // A fresh chunk of the backtrace arena for this workgroup, at least `units` (16-byte units) long: one returning
      // atomic on the arena's bump pointer.  false: the arena is exhausted (chunk_left = 0).  The one place arena space is
      // claimed; what is left of the old chunk is given up.
      auto refill_arena = [&](const uint32_t units) -> bool {
        ColdParams cp = cold_params();
        const uint32_t grab = max(units, cp->chunk_units);
        uint32_t base = WFA_ROW_NONE;
        if (tid == 0) {
          const unsigned long long b = atomicAdd(cp->arena_top, (unsigned long long)grab);
          if (b + grab <= cp->arena_units) base = (uint32_t)b;
        }
        base = block_bcast<NW>(base, bslot);
        chunk_cur = base; chunk_left = (base == WFA_ROW_NONE) ? 0u : grab;
        return base != WFA_ROW_NONE;
      };
      // ---- score 0: M[0][0] = extend(0) ------------------------------------------------------
      // (TILED) the alignment's block of tiles: 64-byte header {marker, wlo, tile columns, budget} + ((budget >> 2) + 1) groups of
      // tile_cols tiles + four tiles of slack (the lanes of a row's last chunk beyond its upper limit store their bytes further right:
      // into tiles of LATER scores of the block, which those scores overwrite, or into the slack)
      uint32_t tile_cols = 0;
      GlobalBytes tiles0 = nullptr;             // byte address of tile (0, 0)
      if constexpr (BT && TILED) {
        tile_cols = (uint32_t)(whi - wlo + 16) >> 4;
        const uint32_t block_units = 4u + 4u * (((uint32_t)budget >> 2) + 1u) * tile_cols + 16u;
        if (chunk_left < block_units + 3u && !refill_arena(block_units + 3u)) status = WFA_ST_NOMEM;
        if (status == WFA_ST_DONE) {
          tab_base = (chunk_cur + 3u) & ~3u;          // (tiles are cache lines: the block starts on one)
          chunk_left -= (tab_base - chunk_cur) + block_units; chunk_cur = tab_base + block_units;
          GlobalBytes const blk = (GlobalBytes)(uintptr_t)cold_params()->arena + (size_t)tab_base * 16u;
          if (tid == 0) {
            uint32_t* const hdr = reinterpret_cast<uint32_t*>(p.arena + (size_t)tab_base * 16u);
            hdr[0] = WFA_ROW_NONE; hdr[1] = (uint32_t)wlo; hdr[2] = tile_cols; hdr[3] = (uint32_t)budget;
          }
          tiles0 = blk + 64;
        }
      } else if constexpr (BT) {
        // the row table (8 bytes per score up to the budget) and the one-cell row of score 0
        const uint32_t tab_units = (uint32_t)(((long long)budget + 2) >> 1);
        if (chunk_left < tab_units + 1 && !refill_arena(tab_units + 1)) status = WFA_ST_NOMEM;
        if (status == WFA_ST_DONE) {
          tab_base = chunk_cur; row_s = chunk_cur + tab_units; chunk_cur += tab_units + 1; chunk_left -= tab_units + 1;
          tab = reinterpret_cast<uint2*>(p.arena + (size_t)tab_base * 16);
          if (tid == 0) p.arena[(size_t)row_s * 16] = 0;     // (its row-table entry: tab_set(0, ...) below)
        }
      }
      // byte address of the tile row of score sc (TILED): tile (sc >> 2, 0), in-tile row sc & 3; and of a diagonal inside it
      auto tile_row = [&](const int sc) -> GlobalBytes { return tiles0 + ((size_t)((uint32_t)sc >> 2) * tile_cols * 64u + ((uint32_t)sc & 3u) * 16u); };
      auto tile_off = [&](const int kk) -> uint32_t { const uint32_t d = (uint32_t)(kk - wlo); return d + 3u * (d & ~15u); };
      uint32_t d0 = 0;
      if (tid == 0) {
        // longest common prefix from (0,0): both sequences start word-aligned
        constexpr int SH = RAW ? 2 : 4, PER = 1 << SH, BITS = RAW ? 3 : 1;
        int h0 = 0, rem = min(plen, tlen);
        while (rem > 0) {
          const uint32_t d = Pw[h0 >> SH] ^ Tw[h0 >> SH];
          const int n = min((int)((d ? (uint32_t)__builtin_ctz(d) : 32u) >> BITS), rem);
          h0 += n; rem -= n;
          if (n < PER) break;
        }
        Mr[kidx0 + GZ] = (OffT)h0;      // (banded: the row of score 0 has lo = 0)
        d0 = ((kend == 0 && h0 >= tlen) ? 1u : 0u) | (h0 >= min(plen, tlen) ? 2u : 0u);
      }
      book.set(0, pack_range(0, 0), ROW_NONE_A, ROW_NONE_A);
      d0 = block_bcast<NW>(d0, bslot);
      done = (d0 & 1u) != 0;
      const bool touched_at_0 = (d0 & 2u) != 0;
      block_sync<NW>();

      // Ring state of score s, as row pointers that advance by one row per score (no multiply, no modulo
      // in the loop): the M rows of s, s-x, s-(o+e) and the I rows of s, s-e (the D ring sits de rows
      // behind the I ring).  In exact mode the pointers address diagonal 0 of their row.  The row book is
      // indexed by (score & bkm); its entries for scores < 0 still hold the "no wavefront" reset value.
      OffT* const m_first = Mr + kidx0;
      OffT* const m_end = m_first + dm * rs;
      OffT* const i_first = m_end;
      OffT* const i_end = i_first + de * rs;
      // the D row that belongs to an I row (same slot of the other ring)
      OffT* const d_first = HYBRID ? Dg + kidx0 : i_first + de * rs;
      auto d_of = [&](OffT* ip) -> OffT* { return d_first + (ip - i_first); };
      OffT* p_m = m_first; OffT* p_x = m_first + (dm - x) * rs; OffT* p_oe = m_first + (dm - oe) * rs;
      OffT* p_ic = i_first; OffT* p_ip = i_first + (de - e) * rs;
      // From a cell of score s on diagonal k the end is at least |k - kend| more gap bases away (an I or
      // D cell may sit inside the gap that is already open, so no opening cost can be assumed), so within
      // the budget only |k - kend| <= (budget - s) / e can still matter (exact, same argument as the
      // window).  [rlo, rhi] is that interval; it loses a diagonal on each side whenever the quotient
      // drops, tracked through the remainder reach_r without a division per score.
      const bool bounded = budget < INT_MAX / 2;
      int reach_r = bounded ? budget % e : INT_MAX;
      int rlo = bounded ? kend - budget / e : INT_MIN / 2, rhi = bounded ? kend + budget / e : INT_MAX / 2;
      // Number of consecutive scores up to s-1 whose wavefront exists with all three components and
      // untrimmed limits.  Once that covers every row the recurrences read, the limits follow from the M
      // limits alone and none of the "no wavefront" cases of wavefront_compute.c:41-71 can occur.
      int regular = 0;
      // Backtrace row of `width` origin bytes for the current score: bump allocation from the block's
      // arena chunk, refilled with one atomic when it runs dry.  false: arena exhausted.
      auto alloc_row = [&](int width) -> bool {
        const uint32_t need = ((uint32_t)width + 15u) >> 4;
        if (need + WFA_ARENA_ROW_SLACK > chunk_left && !refill_arena(need + WFA_ARENA_ROW_SLACK)) return false;
        row_s = chunk_cur; chunk_cur += need; chunk_left -= need;
        return true;
      };
      // Row-table entry of score s.  One wavefront: entries collect in two VGPRs (lane = score & 63) and go out
      // as one coalesced 512-byte store every 64 scores (and at the end of the alignment) instead of one
      // single-lane store + address arithmetic per score.
      int tabv_row = 0, tabv_lo = 0, tab_group = 0;     // the buffered entries belong to scores [64 * tab_group, 64 * tab_group + 63]
      auto tab_set = [&](int score, uint32_t row, int lo_) {
        if constexpr (NW == 1) {
          if ((score >> 6) != tab_group) {
            // (scores without a wavefront leave their lanes stale: such entries are never read)
            tab[(tab_group << 6) + lane] = make_uint2((uint32_t)tabv_row, (uint32_t)tabv_lo);
            tab_group = score >> 6;
          }
          const bool mine = lane == (score & 63);
          tabv_row = mine ? (int)row : tabv_row; tabv_lo = mine ? lo_ : tabv_lo;
        } else {
          if (tid == 0) tab[score] = make_uint2(row, (uint32_t)lo_);
        }
      };
      auto tab_flush = [&](int score) {       // the last group, up to the final score
        if constexpr (NW == 1) {
          if (lane <= (score & 63)) tab[(score & ~63) + lane] = make_uint2((uint32_t)tabv_row, (uint32_t)tabv_lo);
        }
      };
      if constexpr (BT && !TILED) { if (status == WFA_ST_DONE) tab_set(0, row_s, 0); }
      #include "align/cells_generic.inc"
      #include "align/cells_hot.inc"
      #include "align/band_window.inc"
      // Limits of the last score (the lean path derives the next ones from them alone).
      int last_lo = 0, last_hi = 0;
      // Has any M cell reached the end of a sequence (offset == min(plen + k, tlen)) so far?  Only after that can an
      // I or D value run past a sequence end (every such value has a predecessor chain that starts at an M cell sitting
      // on the border), and only such a cell can be the final one.  Until then the lean path needs neither the
      // per-cell overrun tests nor the per-score termination read.
      bool touched_ever = touched_at_0 || cold_params()->no_lean != 0;    // (WFAGPU_NO_LEAN: the lean path is never entered)
      // ---- score loop ----------------------------------------------------------------------------
      if (!done && status == WFA_ST_DONE) for (;;) {
        #include "align/loop_banded.inc"
        #include "align/loop_lean_e1.inc"
        #include "align/loop_lean_any.inc"
        #include "align/loop_lean_hbm.inc"
        #include "align/loop_careful.inc"
      }
      if constexpr (BT) {
        if (status == WFA_ST_DONE) {
          if constexpr (!TILED) tab_flush(s);
          // (the pair's block of tiles, or its row table: wfa_device.h)
          if (tid == 0) cold_params()->bt_final_row[pair] = tab_base;
        }
      }
    }
    if (tid == 0) {
      ColdParams cp = cold_params();
      cp->score[pair] = (status == WFA_ST_DONE) ? s : -1;
      cp->status[pair] = status;
      if (cp->cells) cp->cells[pair] = ncells;
    }
    blk_cells += ncells;
    block_sync<NW>();
  }
  if (tid == 0 && blk_cells) {
    unsigned long long* lc = cold_params()->launch_cells;
    if (lc) atomicAdd(lc, blk_cells);
  }
}

template <int NW, bool BT, typename OffT, bool GR, bool RAW, bool BANDED, bool HYBRID = false, int WPE = 8, bool TIMED = false>
void launch_inst(const WfaAlignParams& p, size_t lds, int grid, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  auto k = wfa_align_kernel<NW, BT, OffT, GR, RAW, BANDED, HYBRID, WPE, TIMED>;
  // the opt-in for large dynamic LDS is sticky per device and per kernel: pay the driver call once
  static thread_local size_t allowed[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > allowed[dev & 15]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    allowed[dev & 15] = lds;
  }
  wfa_launch_timed(k, dim3(grid), dim3(NW * 64), lds, stream, ev0, ev1, p);
}

template <int NW, bool BT, typename OffT, bool GR, bool RAW, bool BANDED, bool HYBRID = false, int WPE = 8>
int occ_inst(size_t lds) {
  auto k = wfa_align_kernel<NW, BT, OffT, GR, RAW, BANDED, HYBRID, WPE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), NW * 64, lds) != hipSuccess) {
    nb = 0;
    (void)hipGetLastError();      // (a plan that does not fit is an answer, not an error the next launch should trip over)
  }
  return nb;
}

// tier -> instantiation (banded kernels exist for the packed LDS tiers only)
template <bool BT, bool RAW>
void launch_tier(const WfaAlignParams& p, int tier, size_t lds, int grid, hipStream_t stream, int wpe, hipEvent_t ev0, hipEvent_t ev1) {
  switch (tier) {
    case 0:
      // (the byte-compare class keeps the one instantiation: it is the rare path)
      if constexpr (!RAW) {
        if (wpe == 7) { launch_inst<1, BT, int16_t, false, false, false, false, 7>(p, lds, grid, stream, ev0, ev1); break; }
        if (wpe == 6) { launch_inst<1, BT, int16_t, false, false, false, false, 6>(p, lds, grid, stream, ev0, ev1); break; }
        if (wpe == 4) { launch_inst<1, BT, int16_t, false, false, false, false, 4>(p, lds, grid, stream, ev0, ev1); break; }
      }
      launch_inst<1, BT, int16_t, false, RAW, false>(p, lds, grid, stream, ev0, ev1); break;
    case 1:
      if constexpr (BT && !RAW) { if (p.dbg_times) { launch_inst<4, true, int16_t, false, false, false, false, 8, true>(p, lds, grid, stream, ev0, ev1); break; } }
      launch_inst<4, BT, int16_t, false, RAW, false>(p, lds, grid, stream, ev0, ev1); break;
    case 2: launch_inst<16, BT, int16_t, false, RAW, false>(p, lds, grid, stream, ev0, ev1); break;
    case 6:      // two waves per alignment (exact: A/B hook tuning.exact_two_waves; the byte-compare class keeps four)
      if constexpr (!RAW) launch_inst<2, BT, int16_t, false, false, false>(p, lds, grid, stream, ev0, ev1);
      else launch_inst<4, BT, int16_t, false, RAW, false>(p, lds, grid, stream, ev0, ev1);
      break;
    case 4:
      if constexpr (BT && !RAW) { if (p.dbg_times) { launch_inst<16, true, int16_t, false, false, false, true, 8, true>(p, lds, grid, stream, ev0, ev1); break; } }
      if constexpr (!RAW) launch_inst<16, BT, int16_t, false, false, false, true>(p, lds, grid, stream, ev0, ev1);
      break;    // hybrid ring
    default:
      if (p.ring16) launch_inst<16, BT, int16_t, true, RAW, false>(p, lds, grid, stream, ev0, ev1);
      else launch_inst<16, BT, int32_t, true, RAW, false>(p, lds, grid, stream, ev0, ev1);
      break;
  }
}
// (tier 6: TWO waves per alignment, banded kernels only -- a band of 512 diagonals on many pairs: twice the wavefronts per CU of the
// one-wave tier at the same LDS, four chunks per wave and score to spread the per-score part over)
template <bool BT>
void launch_tier_banded(const WfaAlignParams& p, int tier, size_t lds, int grid, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  switch (tier) {
    case 0: launch_inst<1, BT, int16_t, false, false, true>(p, lds, grid, stream, ev0, ev1); break;
    case 6: launch_inst<2, BT, int16_t, false, false, true>(p, lds, grid, stream, ev0, ev1); break;
    case 1: launch_inst<4, BT, int16_t, false, false, true>(p, lds, grid, stream, ev0, ev1); break;
    default: launch_inst<16, BT, int16_t, false, false, true>(p, lds, grid, stream, ev0, ev1); break;
  }
}
template <bool BT, bool RAW>
int occ_tier(int tier, size_t lds, int wpe) {
  switch (tier) {
    case 0:
      if constexpr (!RAW) {
        if (wpe == 7) return occ_inst<1, BT, int16_t, false, false, false, false, 7>(lds);
        if (wpe == 6) return occ_inst<1, BT, int16_t, false, false, false, false, 6>(lds);
        if (wpe == 4) return occ_inst<1, BT, int16_t, false, false, false, false, 4>(lds);
      }
      return occ_inst<1, BT, int16_t, false, RAW, false>(lds);
    case 1: return occ_inst<4, BT, int16_t, false, RAW, false>(lds);
    case 2: return occ_inst<16, BT, int16_t, false, RAW, false>(lds);
    case 6: if constexpr (!RAW) return occ_inst<2, BT, int16_t, false, false, false>(lds); else return occ_inst<4, BT, int16_t, false, RAW, false>(lds);
    case 4: if constexpr (!RAW) return occ_inst<16, BT, int16_t, false, false, false, true>(lds); else return 0;
    default: return occ_inst<16, BT, int32_t, true, RAW, false>(lds);
  }
}
template <bool BT>
int occ_tier_banded(int tier, size_t lds) {
  switch (tier) {
    case 0: return occ_inst<1, BT, int16_t, false, false, true>(lds);
    case 6: return occ_inst<2, BT, int16_t, false, false, true>(lds);
    case 1: return occ_inst<4, BT, int16_t, false, false, true>(lds);
    default: return occ_inst<16, BT, int16_t, false, false, true>(lds);
  }
}

}  // namespace

size_t wfa_align_lds_bytes(const WfaAlignParams& p, int tier) {
  size_t ring = 0;
  if (tier == 4) ring = (((size_t)(p.dm + p.de) * p.rs * sizeof(int16_t)) + 15) & ~(size_t)15;       // hybrid: M and I rings
  else if (tier != 3) ring = (((size_t)(p.dm + 2 * p.de + ((tier == 0 && p.band_width <= 0) ? 1 : 0)) * p.rs * sizeof(int16_t)) + 15) & ~(size_t)15;     // (+ the run-limit row of the one-wave exact kernels)
  const size_t seq = (size_t)2 * p.seq_words_cap * 4;
  // reduction slots [24] + broadcast [2] + (NW > 1) row book [3][book]
  const size_t bk = (size_t)p.book_mask + 1;
  const size_t meta = (size_t)(24 + 2 + ((tier == 0 || p.band_width > 0) ? 0 : 3 * bk)) * 4;     // (tiers 1, 2, 3, 4 of the exact search: row book in LDS)
  return ((ring + seq + meta) + 15) & ~(size_t)15;
}

void wfa_launch_align(const WfaAlignParams& p, int tier, bool with_bt, bool raw, int grid, hipStream_t stream, int wpe, hipEvent_t ev0, hipEvent_t ev1) {
  const size_t lds = wfa_align_lds_bytes(p, tier);
  if (p.band_width > 0) {
    if (with_bt) launch_tier_banded<true>(p, tier, lds, grid, stream, ev0, ev1); else launch_tier_banded<false>(p, tier, lds, grid, stream, ev0, ev1);
    return;
  }
  if (with_bt) { if (raw) launch_tier<true, true>(p, tier, lds, grid, stream, wpe, ev0, ev1); else launch_tier<true, false>(p, tier, lds, grid, stream, wpe, ev0, ev1); }
  else { if (raw) launch_tier<false, true>(p, tier, lds, grid, stream, wpe, ev0, ev1); else launch_tier<false, false>(p, tier, lds, grid, stream, wpe, ev0, ev1); }
}

int wfa_align_max_blocks_per_cu(int tier, bool with_bt, bool raw, bool banded, size_t lds, int wpe) {
  if (banded) return with_bt ? occ_tier_banded<true>(tier, lds) : occ_tier_banded<false>(tier, lds);
  if (with_bt) return raw ? occ_tier<true, true>(tier, lds, wpe) : occ_tier<true, false>(tier, lds, wpe);
  return raw ? occ_tier<false, true>(tier, lds, wpe) : occ_tier<false, false>(tier, lds, wpe);
}

// Loads this translation unit's code object on the current device (the runtime loads a code object at the first launch of
// any of its kernels: 5-25 ms each): launch_alignments* call it while a cold call waits for its first upload.
namespace { __global__ void k_prime_align() {} }
void wfa_prime_align(hipStream_t stream) { hipLaunchKernelGGL(k_prime_align, dim3(1), dim3(64), 0, stream); }
