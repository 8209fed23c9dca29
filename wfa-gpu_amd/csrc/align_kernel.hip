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
//     has closed-form limits (e == 1), in-place state, one loop exit, and touches the row book only when it leaves.
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
//   * For CIGARs each cell emits ONE origin byte (64 consecutive bytes per wavefront store) into a
//     bump-allocated arena; a per-alignment row table (8 bytes per score) locates the rows.  No
//     O(max_error^2) per-alignment reservation and nothing to memset between alignments.
#include <type_traits>

#include "wfa_device.h"

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
  RowBook<NW> book;
  book.reset();
  if constexpr (NW == 1) { book.A = book.I = book.D = nullptr; }
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
      for (int i = tid; i < pwords; i += NT) Pw[i] = gp[i];
      for (int i = tid; i < twords; i += NT) Tw[i] = gt[i];
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
      if constexpr (NW == 1) book.reset();
      else {
        for (int i = tid; i <= bkm; i += NT) { book.A[i] = book.I[i] = book.D[i] = ROW_NONE_A; }
        if (tid < 24) red[tid] = (tid & 7) == 6 ? 0 : (((tid & 7) & 1) ? INT_MIN : INT_MAX);
      }
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
      if constexpr (BT) {
        // the row table (8 bytes per score up to the budget) and the one-cell row of score 0
        const uint32_t tab_units = (uint32_t)(((long long)budget + 2) >> 1);
        if (chunk_left < tab_units + 1 && !refill_arena(tab_units + 1)) status = WFA_ST_NOMEM;
        if (status == WFA_ST_DONE) {
          tab_base = chunk_cur; row_s = chunk_cur + tab_units; chunk_cur += tab_units + 1; chunk_left -= tab_units + 1;
          tab = reinterpret_cast<uint2*>(p.arena + (size_t)tab_base * 16);
          if (tid == 0) p.arena[(size_t)row_s * 16] = 0;     // (its row-table entry: tab_set(0, ...) below)
        }
      }
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
      if constexpr (BT) { if (status == WFA_ST_DONE) tab_set(0, row_s, 0); }
      // ---- the cells of one score.  Lanes past the end recompute cell `hi` (same values, same
      // addresses), so no store needs an exec mask.  The vector ALU is the unit this kernel saturates (one
      // integer wave64 instruction holds its SIMD for 4 cycles), so everything uniform is folded into scalar
      // row bases: each LDS address is one v_lshl_add of the diagonal.
      //   rb_mx[k] = M[s-x][k]   rb_mo[k] = M[s-o-e][k-1], rb_mo[k+2] = M[s-o-e][k+1]
      //   rb_ie[k] = I[s-e][k-1] rb_de[k] = D[s-e][k+1]     wb_*[k]: the rows written now
      // LEAN (regular regime, no cell has touched a sequence end yet): no value can run past an end, so the
      // overrun test and the saturation of I are dropped.  wave_touch: lanes whose M cell reached min(plen + k, tlen).
      // (BANDED: wx, wo, we = the windows -- pack_range(lo, hi), ROW_NONE_A for a row that does not exist -- of the rows M[s-x],
      // M[s-o-e] and I/D[s-e]: every read is range-checked against its row's window like the reference's get_offset,
      // lib/kernels/sequence_distance_kernel_aband.cu:28-33, because after a re-centring the rows of neighbouring scores
      // can lie anywhere relative to each other)
      auto cells_of_score = [&](auto lean_tag, const int lo, const int hi, uint8_t* codes, const OffT* rb_mx, const OffT* rb_mo,
                                const OffT* rb_ie, const OffT* rb_de, OffT* wb_m, OffT* wb_i, OffT* wb_d,
                                bool& my_over, unsigned long long& wave_touch, const int wx = 0, const int wo = 0, const int we = 0) {
        constexpr bool LEAN = decltype(lean_tag)::value;
        // (the 16-bit LDS tiers never get here with LEAN set: their lean loops call hot_cells below)
        static_assert(!(LEAN && HOT), "lean cells of the LDS tiers live in hot_cells");
        for (int k0 = lo; k0 <= hi; k0 += NT) {
          const int k = min(k0 + tid, hi);
          // recurrences (wavefront_compute_affine.c:66-84)
          int ins, del, mv0;
          uint32_t code = 0;
          {
            int m_x, m_ol, m_or, i_e, d_e;
            if constexpr (BANDED) {
              auto rd = [&](const OffT* base, const int idx, const int kk, const int w) -> int {
                return (kk >= range_lo(w) && kk <= range_hi(w)) ? (int)base[idx] : (int)OffNull<OffT>::value;
              };
              m_x = rd(rb_mx, k, k, wx);
              m_ol = rd(rb_mo, k, k - 1, wo);
              m_or = rd(rb_mo, k + 2, k + 1, wo);
              i_e = rd(rb_ie, k, k - 1, we);
              d_e = rd(rb_de, k, k + 1, we);
            } else {
              m_x = (int)rb_mx[k];
              m_ol = (int)rb_mo[k];
              m_or = (int)rb_mo[k + 2];
              i_e = (int)rb_ie[k];
              d_e = (int)rb_de[k];
            }
            ins = max(m_ol, i_e) + 1;
            del = max(m_or, d_e);
            const int mis = m_x + 1;
            mv0 = max(del, max(mis, ins));
            if constexpr (BT) {
              // tie-breaks: gap extension wins over gap open on equal offsets
              // (wavefront_compute_affine.c:135-143,153-161); for M: mismatch, then deletion, then
              // insertion (wavefront_backtrace.c:48-59)
              // (the M origin of a cell that is not valid is never read: the backtrace only visits valid cells)
              code = (i_e >= m_ol ? BT_I_EXT : 0u) | (d_e >= m_or ? BT_D_EXT : 0u);
              code |= (mis == mv0) ? BT_M_X : ((del == mv0) ? BT_M_D : BT_M_I);
            }
          }
          // !(h > tlen || v > plen), unsigned so that negatives fail too
          const bool ok = ((unsigned)mv0 <= (unsigned)tlen) & ((unsigned)(mv0 - k) <= (unsigned)plen);
          // An I (D) value can only be out of range by running past the text (pattern) end; such
          // values are rare (last scores only) and send the row through the exact trimming pass
          // below.  Everywhere else "invalid" means negative, which already reads as NULL, so the
          // computed limits can stand in for the trimmed ones.
          if constexpr (!LEAN) my_over |= (ins > tlen) || (del - k > plen);
          // extend = longest common prefix from (v,h) (wavefront_extend.c:174-199), PER symbols per step.  No exec
          // mask for the cells that are not valid: a wave instruction costs the same with any lane active, mask
          // bookkeeping costs scalar instructions, and LDS reads at their meaningless addresses are harmless (an
          // address beyond the workgroup's allocation reads as 0); their result is dropped below.  The first step is
          // straight-line code (most cells stop inside their first word); only when some lane matched a whole word
          // with more to go does the wave enter the loop, in which lanes that are done carry left == 0 and idle along.
          int h = mv0;
          constexpr int SH = RAW ? 2 : 4, PER = 1 << SH, BITS = RAW ? 3 : 1;
          // the run cannot pass either sequence end: h <= tlen and v = h - k <= plen
          const int hmax = min(plen + k, tlen);
          {
            const int v = mv0 - k;
            const int rem = hmax - h;
            // word pointers and bit offsets are fixed for the whole run: a lane that goes on has
            // consumed exactly PER symbols = one word
            const char* pp = reinterpret_cast<const char*>(Pw + (v >> SH));
            const char* tp = reinterpret_cast<const char*>(Tw + (h >> SH));
            const uint32_t sa = (uint32_t)v << BITS, sb = (uint32_t)h << BITS;
            uint32_t fb, d0w;
            {
              const uint32_t* pw = reinterpret_cast<const uint32_t*>(pp);
              const uint32_t* tw = reinterpret_cast<const uint32_t*>(tp);
              d0w = __builtin_amdgcn_alignbit(pw[1], pw[0], sa) ^ __builtin_amdgcn_alignbit(tw[1], tw[0], sb);
              // v_ffbl_b32 returns 0xFFFFFFFF for 0, so "all equal" is a huge positive count
              asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(d0w));
            }
            {
              // (one v_min3_i32: the compiler splits the two mins into an unsigned and a signed one)
              int adv;
              asm("v_min3_i32 %0, %1, %2, %3" : "=v"(adv) : "v"((int)(fb >> BITS)), "v"(rem), "n"(PER));
              h += adv;
            }
            // the whole word matched: the run may go on (if anything remains)
            const bool more = ok & (d0w == 0u);
            if (__builtin_amdgcn_ballot_w64(more) != 0ull) {
              int left = more ? max(rem - PER, 0) : 0;
              while (__builtin_amdgcn_ballot_w64(left > 0) != 0ull) {
                pp += 4; tp += 4;
                const uint32_t* pw = reinterpret_cast<const uint32_t*>(pp);
                const uint32_t* tw = reinterpret_cast<const uint32_t*>(tp);
                const uint32_t d = __builtin_amdgcn_alignbit(pw[1], pw[0], sa) ^ __builtin_amdgcn_alignbit(tw[1], tw[0], sb);
                asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(d));
                const int nn = min(min((int)(fb >> BITS), PER), left);
                h += nn;
                left = (nn == PER) ? left - PER : 0;
              }
            }
          }
          const int mv = ok ? h : OffNull<OffT>::value;
          wave_touch |= __builtin_amdgcn_ballot_w64(mv == hmax);      // (NULL never equals it)
          // M and D offsets never exceed the text length, so they fit 16 bits as they are; only an I
          // chain running past the text end keeps growing and is saturated by off_store
          wb_m[k] = (OffT)mv;
          if constexpr (LEAN) wb_i[k] = (OffT)ins; else wb_i[k] = off_store<OffT>(ins);
          wb_d[k] = (OffT)del;
          if constexpr (BT) codes[(uint32_t)(k - lo)] = (uint8_t)code;
        }
      };
      // ---- the lean cells of one score for the one-wavefront 16-bit LDS tier: rows given as LDS byte addresses of their
      // diagonal 0 (a_oe: M[s-o-e], a_x: M[s-x], a_ip / a_dp: I / D of s-e, a_m / a_ic / a_dc: the rows written), `codes`:
      // this lane's origin byte in the row of origin bytes (global), wm1 = hi - lo.  Same cells, values and origin bytes
      // as cells_of_score's lean form.
      typedef __attribute__((address_space(3))) OffT* LdsRow;
      typedef __attribute__((address_space(3))) const uint32_t* LdsWords;
      typedef __attribute__((address_space(1))) uint8_t* GlobalBytes;
      typedef __attribute__((address_space(1))) OffT* GlobalRow;
      OffT* const hm_row0 = Mr + (dm + 2 * de) * rs + kidx0;      // (HOT) diagonal 0 of the run-limit row
      auto hot_cells = [&](const int lo, const int wm1, GlobalBytes& codes, const uint32_t a_oe, const uint32_t a_x, const uint32_t a_m,
                           const uint32_t a_ip, const uint32_t a_ic, const uint32_t a_dp, const uint32_t a_dc, const uint32_t pw_addr, const uint32_t tw_addr,
                           const uint32_t a_hm, unsigned long long& touch) {
        constexpr int SH = RAW ? 2 : 4, PER = 1 << SH, BITS = RAW ? 3 : 1;
        // Every role tag carries TB >= PER besides its origin bits (the backtrace masks them off), so that any tagged value
        // that is valid (offset >= 0) is >= PER and v_med3(value, 0, PER) is PER for a valid cell and 0 for a NULL one.
        constexpr uint32_t TB = 16;
        static_assert(TB >= (uint32_t)PER && (TB & (BT_M_MASK | BT_D_EXT | BT_I_EXT)) == 0, "tag base");
        // a group of up to four chunks from kq (this lane's diagonal; the lane base of every row involved is cell kq - 1:
        // all neighbours are immediates from there) and codes; returns whether the row goes on beyond the group
        auto group = [&](const int kq, const GlobalBytes codes, const int left) -> bool {
          bool more_groups;
          const uint32_t vb = (uint32_t)(kq - 1) << 1;
          const int kq16 = kq << 16;
          // (one v_add each, once per score: the empty asm keeps the compiler from re-forming them in every chunk)
          uint32_t q_mo = vb + a_oe, q_mx = vb + a_x, q_wm = vb + a_m, q_ri = vb + a_ip, q_wi = vb + a_ic;
          uint32_t q_rd = vb + a_dp, q_wd = vb + a_dc, q_hm = vb + a_hm;
          asm volatile("" : "+v"(q_mo), "+v"(q_mx), "+v"(q_wm), "+v"(q_ri), "+v"(q_wi), "+v"(q_rd), "+v"(q_wd), "+v"(q_hm));
          const LdsRow r_mo = (LdsRow)q_mo, r_mx = (LdsRow)q_mx, w_m = (LdsRow)q_wm;
          const LdsRow r_i = (LdsRow)q_ri, w_i = (LdsRow)q_wi, r_hm = (LdsRow)q_hm;
          // the D rows: LDS like the others, or (hybrid ring) global memory, where a_dp / a_dc are byte offsets into the ring
          auto d_row = [&](const uint32_t q) {
            if constexpr (HYBRID) return (GlobalRow)((GlobalBytes)(uintptr_t)d_first + (ptrdiff_t)(int32_t)q);      // (diagonals below 0: negative offsets)
            else return (LdsRow)q;
          };
          const auto r_d = d_row(q_rd), w_d = d_row(q_wd);
          auto chunk = [&](auto uc, auto partial_tag, const unsigned long long act) {
            constexpr int O = decltype(uc)::value * NT;
            constexpr bool PARTIAL = decltype(partial_tag)::value;
            const int k = kq + O;
            const uint32_t u_ol = (uint16_t)r_mo[O], u_or = (uint16_t)r_mo[O + 2], u_ie = (uint16_t)r_i[O],
                           u_de = (uint16_t)r_d[O + 2], u_x = (uint16_t)r_mx[O + 1];
            // min(plen + k, tlen): how far a run on this diagonal can go
            const int hmax = HM_ROW ? (int)(uint16_t)r_hm[O + 1] : min(plen + k, tlen);
            int ins_c = max((int)((u_ol << 16) + (0x10000u | TB | BT_M_I)), (int)((u_ie << 16) + (0x10000u | TB | BT_M_I | BT_I_EXT)));
            int del_t = max((int)((u_or << 16) | (TB | BT_M_D)), (int)((u_de << 16) | (TB | BT_M_D | BT_D_EXT)));
            const int mis_c = (int)((u_x << 16) + (0x10000u | TB | BT_M_X));
            const int mv_t = max(del_t, max(mis_c, ins_c));
            uint32_t code = 0;
            if constexpr (BT) {
              uint32_t c1;
              asm("v_bfi_b32 %0, 2, %1, %2" : "=v"(c1) : "v"(del_t), "v"(mv_t));      // bit 1 from the deletion winner
              asm("v_bfi_b32 %0, 1, %1, %2" : "=v"(code) : "v"(ins_c), "v"(c1));      // bit 0 from the insertion winner
            }
            const int mv0 = mv_t >> 16;
            // Nothing has touched a sequence end: "not valid" = NULL = negative.  A NULL cell gets a run length of 0 and is
            // stored as it is -- NULL plus at most one per score, which stays negative for every score 16 bits can hold.
            int cap;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(cap) : "v"(mv_t), "n"(PER));
            int h = mv0;
            {
              const int rem = hmax - h;
              uint32_t pa, ta;      // word addresses: base + 4 * (symbol index / PER)
              uint32_t sa, sb;      // bit offsets of the run's first symbol inside its word (v_alignbit takes the low five bits)
              if constexpr (!RAW) {
                // (gfx950 issues v_add/v_sub/v_and/v_or/v_xor and the RIGHT shifts at one wave instruction per ~2.5 cycles,
                // left shifts, min/max, compares, every three-operand and every SDWA form at one per ~4.2 --
                // profiles/r04/valu_classes.txt: v and h are taken from the tagged value with subtractions and right shifts.
                // The tag occupies bits 4:0 and bit 15 is clear, so (x >> 15) is twice the high half.)
                const int vt = mv_t - kq16 - (O << 16);             // (h - k) << 16 | tag
                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(pa) : "v"(vt >> (16 + SH)), "s"(pw_addr));
                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ta) : "v"(mv_t >> (16 + SH)), "s"(tw_addr));
                sa = (uint32_t)vt >> 15; sb = (uint32_t)mv_t >> 15;
              } else {
                const int v = mv0 - k;
                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(pa) : "v"(v >> SH), "s"(pw_addr));
                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ta) : "v"(h >> SH), "s"(tw_addr));
                sa = (uint32_t)v << BITS; sb = (uint32_t)h << BITS;
              }
              uint32_t fb;
              {
                const LdsWords pw = (LdsWords)pa; const LdsWords tw = (LdsWords)ta;
                const uint32_t d0w = __builtin_amdgcn_alignbit(pw[1], pw[0], sa) ^ __builtin_amdgcn_alignbit(tw[1], tw[0], sb);
                asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(d0w));       // (0xFFFFFFFF for 0: "all equal" is a huge count)
              }
              int adv;
              asm("v_min3_i32 %0, %1, %2, %3" : "=v"(adv) : "v"((int)(fb >> BITS)), "v"(rem), "v"(cap));
              h += adv;
              // a whole word matched (a valid cell with at least a word to go): the run may go on
              const bool more = adv == PER;
              if (__builtin_amdgcn_ballot_w64(more) != 0ull) {
                int togo = more ? max(rem - PER, 0) : 0;
                while (__builtin_amdgcn_ballot_w64(togo > 0) != 0ull) {
                  pa += 4; ta += 4;
                  const LdsWords pw = (LdsWords)pa; const LdsWords tw = (LdsWords)ta;
                  const uint32_t d = __builtin_amdgcn_alignbit(pw[1], pw[0], sa) ^ __builtin_amdgcn_alignbit(tw[1], tw[0], sb);
                  asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(d));
                  const int nn = min(min((int)(fb >> BITS), PER), togo);
                  h += nn;
                  togo = (nn == PER) ? togo - PER : 0;
                }
              }
            }
            int mv = h;
            if constexpr (PARTIAL) {
              const bool active = __builtin_amdgcn_inverse_ballot_w64(act);
              mv = active ? h : OffNull<OffT>::value;
              ins_c = active ? ins_c : (int)0x80000000u; del_t = active ? del_t : (int)0x80000000u;
            }
            touch |= __builtin_amdgcn_ballot_w64(mv == hmax);      // (a NULL never equals it)
            w_m[O + 1] = (OffT)mv;
            w_i[O + 1] = (OffT)(ins_c >> 16);     // (high halves: ds_write_b16_d16_hi, no unpacking)
            w_d[O + 1] = (OffT)(del_t >> 16);
            if constexpr (BT) codes[O] = (uint8_t)code;
          };
          // chunks 0..3 of the group, nested so that every decision is one scalar compare and branch
          more_groups = false;
          auto from = [&](auto&& self, auto uc) -> void {
            constexpr int Uc = decltype(uc)::value;
            const int n_act = left - Uc * NT;       // cells of the row from this wave's chunk on (wave-uniform)
            if (n_act >= 64) {
              chunk(uc, std::false_type{}, 0ull);
              if constexpr (Uc < 3) { if (n_act > NT) self(self, std::integral_constant<int, Uc + 1>{}); }
              else more_groups = n_act > NT;
            } else if (NW == 1 || n_act > 0) {
              chunk(uc, std::true_type{}, (1ull << n_act) - 1ull);
            }
          };
          from(from, std::integral_constant<int, 0>{});
          return more_groups;
        };
        // (several waves: each one's count of remaining cells starts at its own first diagonal)
        const int left0 = wm1 + 1 - ((NW == 1) ? 0 : __builtin_amdgcn_readfirstlane(stid & ~63));
        // (`codes` is advanced in place and put back: a second copy of the 64-bit address would not fit the registers)
        int kq = stid + lo, left = left0;
        if constexpr (NW == 1) {
          if (__builtin_expect(group(kq, codes, left), 0)) {
            // (wider than four chunks: rare in this tier, kept out of the way of the common case)
            uint32_t adv = 0;
            do { kq += 4 * NT; codes += 4 * NT; adv += 4u * NT; left -= 4 * NT; } while (group(kq, codes, left));
            codes -= adv;
          }
        } else {
          uint32_t adv = 0;
          while (group(kq, codes, left)) { kq += 4 * NT; codes += 4 * NT; adv += 4u * NT; left -= 4 * NT; }
          codes -= adv;
        }
      };
      // Adaptive band: the window of a score, by the REFERENCE's rule (lib/kernels/sequence_distance_kernel_aband.cu:91-130, the
      // CIGAR kernel sequence_alignment_kernel_aband.cu:147-205 alike; restated on the CPU in oracle/band_oracle.c, which the tests
      // compare this kernel's banded scores with, pair by pair):
      //   hi = max(Mx.hi, max(Mo.hi, I.hi, D.hi) + 1), lo = min(Mx.lo, min(Mo.lo, I.lo, D.lo) - 1) over the windows of the four input
      //   rows (a row that does not exist yet counts as the window [0, 0] of an empty slot, :262-281);
      //   too wide: hi--, lo++ in turn until beta diagonals are left (:100-104);
      //   when the MISMATCH-source row M[s-x] is full width and s % lambda == 0: the diagonal of M[s-x] in [lo, hi) -- its last one is
      //   not looked at -- whose offset is closest to the end (max(plen - v, tlen - h), first minimum) becomes the centre:
      //   lo = centre - beta / 2, hi = lo + beta - 1, unconditionally (:114-130).
      // No clipping to the sequence ends or to the score budget (the reference has none); deterministic where the reference's kernels
      // race (a window is read before thread 0 has published it).  Every thread calls it with the same arguments.
      // (round 3 shipped a rule of its own here -- re-centring only when the wavefront overflowed, clamped into the old window,
      // at most two diagonals per score -- which lost pairs the reference's rule keeps: profiles/r04/banded.md.)
      auto band_window = [&](int& lo, int& hi, const bool period_start, const int wx, const int wo, const int we, const OffT* row_mx) {
        const int beta = p.band_width;
        hi = max(range_hi(wx), max(range_hi(wo), range_hi(we)) + 1);
        lo = min(range_lo(wx), min(range_lo(wo), range_lo(we)) - 1);
        const int excess = (hi - lo + 1) - beta;
        if (excess > 0) { hi -= (excess + 1) / 2; lo += excess / 2; }
        const int mxlo = range_lo(wx), mxhi = range_hi(wx);
        if (mxhi - mxlo >= beta - 1 && period_start) {      // (period_start: score % band_period == 0)
          uint32_t best = 0xFFFFFFFFu;
          for (int kk = mxlo + tid; kk < mxhi; kk += NT) {
            const int off = (int)row_mx[kk];
            if (off >= 0) {
              const int dist = max(plen - (off - kk), tlen - off);
              best = min(best, ((uint32_t)dist << 16) | (uint32_t)(kk - mxlo));
            }
          }
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, d));
          if constexpr (NW > 1) {
            if (tid == 0) bslot[0] = 0xFFFFFFFFu;
            __syncthreads();
            if (lane == 0) atomicMin(&bslot[0], best);
            __syncthreads();
            best = bslot[0];
            __syncthreads();
          }
          best = __builtin_amdgcn_readfirstlane(best);      // (uniform: keeps what follows on the scalar unit)
          const int centre = mxlo + (best != 0xFFFFFFFFu ? (int)(best & 0xFFFFu) : 0);
          lo = centre - beta / 2; hi = lo + beta - 1;
        }
      };
      // Limits of the last score (the lean path derives the next ones from them alone).
      int last_lo = 0, last_hi = 0;
      // Has any M cell reached the end of a sequence (offset == min(plen + k, tlen)) so far?  Only after that can an
      // I or D value run past a sequence end (every such value has a predecessor chain that starts at an M cell sitting
      // on the border), and only such a cell can be the final one.  Until then the lean path needs neither the
      // per-cell overrun tests nor the per-score termination read.
      bool touched_ever = touched_at_0 || cold_params()->no_lean != 0;    // (WFAGPU_NO_LEAN: the lean path is never entered)
      // ---- score loop ----------------------------------------------------------------------------
      if (!done && status == WFA_ST_DONE) for (;;) {
        // ---- the whole banded search: one loop, the reference's windows (band_window), the shared cells.  A ring row holds the
        // diagonals of its window relative to the window's lower limit (column GZ = diagonal lo) between two NULL guard zones of
        // GZ columns.  While every input row lies within GZ - 1 diagonals of the new window -- always, but for the few scores after
        // a re-centring jump -- the lean cells read them without any range test (a read beside a row lands in its guard zones);
        // otherwise, and once a cell has touched a sequence end, the generic cells run with every read checked against its row's
        // window.  Scores without a wavefront and M-only scores (before the first gap can exist) follow the reference's
        // existence logic (sequence_distance_kernel_aband.cu:331-372); a row that does not exist is an all-NULL slot with the
        // window [0, 0], which is what the reference's never-written slots are (for penalty sets with e > 1 the reference re-reads
        // whatever an older score left in such a slot: not reproduced -- INTEGRATION.md).
        if constexpr (BANDED) {
          constexpr int W00 = 0;                                      // pack_range(0, 0)
          unsigned long long ex_m = 1ull, ex_i = 0ull;                // bit (score & 63): M / the gap components of that score exist
          int phase = 0;                                              // score % band_period, kept by counting (no division per score)
          int full_rows = 0;                                          // consecutive scores so far whose rows were band_width wide
          if constexpr (NW > 1) { if (tid == 0) bslot[1] = 0u; }
          // (row book: M windows; rows that do not exist -- negative scores included -- read as the window [0, 0])
          if constexpr (NW == 1) { book.a = W00; } else { for (int i = tid; i <= bkm; i += NT) book.A[i] = W00; }
          book.set_a(0, W00);
          block_sync<NW>();
          for (;;) {
            const int ns = s + 1;
            if (ns > budget) { status = WFA_ST_SCORE; break; }
            if (++phase == p.band_period) phase = 0;
            const bool has_oe = ns >= oe;
            const bool e_oe = has_oe && ((ex_m >> ((ns - oe) & 63)) & 1ull), e_ie = has_oe && ((ex_i >> ((ns - e) & 63)) & 1ull);
            const bool e_x = ns >= x && ((ex_m >> ((ns - x) & 63)) & 1ull);
            const bool gap = e_oe || e_ie, mex = gap || e_x;
            const unsigned long long bit = 1ull << (ns & 63);
            ex_m = mex ? (ex_m | bit) : (ex_m & ~bit);
            ex_i = gap ? (ex_i | bit) : (ex_i & ~bit);
            s = ns;
            p_m += rs;  if (p_m == m_end) p_m = m_first;
            p_x += rs;  if (p_x == m_end) p_x = m_first;
            p_oe += rs; if (p_oe == m_end) p_oe = m_first;
            p_ic += rs; if (p_ic == i_end) p_ic = i_first;
            p_ip += rs; if (p_ip == i_end) p_ip = i_first;
            OffT* const out_m = p_m; OffT* const out_i = p_ic; OffT* const out_d = d_of(p_ic);
            if (!mex) {
              // no wavefront at this score: an all-NULL slot
              for (int q = tid; q < p.band_width + 2 * GZ; q += NT) {
                out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
              }
              book.set_a(s & bkm, W00);
              block_sync<NW>();
              continue;
            }
            // windows of the input rows ([0, 0] where a row does not exist)
            const int wx = ns >= x ? book.get_a((ns - x) & bkm) : W00;
            const int wo = has_oe ? book.get_a((ns - oe) & bkm) : W00;
            const int we = e_ie ? book.get_a((ns - e) & bkm) : W00;
            int lo, hi;
            if (gap) band_window(lo, hi, phase == 0, wx, wo, we, p_x + (GZ - range_lo(wx)));
            else { lo = range_lo(wx); hi = range_hi(wx); }            // M only: the window of M[s-x] (:54-74)
            const int width = hi - lo + 1;
            ncells += (uint32_t)width;
            uint8_t* codes = nullptr;
            if constexpr (BT) {
              if (!alloc_row(width)) { status = WFA_ST_NOMEM; break; }
              tab_set(s, row_s, lo);
              codes = p.arena + (size_t)row_s * 16;
            }
            // ring invariant of relative rows: NULL guard zones on both sides of the row, whatever the slot held before.  (Once
            // every slot of the ring has held a full-width row, the zones ARE NULL -- nothing but NULLs is ever stored beside a
            // row -- and stay so while the rows stay full width.)
            full_rows = width == p.band_width ? full_rows + 1 : 0;
            if (full_rows <= dm + 1) {
              for (int j = tid; j < 2 * GZ; j += NT) {
                const int q = j < GZ ? j : width + j;
                out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
              }
            }
            // rows that exist: their own mapping (column GZ = their lower limit); others: all NULL, any mapping
            const int rel_cur = GZ - lo;
            const int rel_x = e_x ? GZ - range_lo(wx) : rel_cur, rel_oe = e_oe ? GZ - range_lo(wo) : rel_cur, rel_e = e_ie ? GZ - range_lo(we) : rel_cur;
            auto near = [&](const bool exists, const int w) { return !exists || (range_lo(w) - lo <= GZ - 1 && hi - range_hi(w) <= GZ - 1); };
            unsigned long long touch = 0;
            bool lean_cells = false;
            if constexpr (HOT) lean_cells = !touched_ever && near(e_x, wx) && near(e_oe, wo) && near(e_ie, we);
            if (lean_cells) {
              if constexpr (HOT) {
                GlobalBytes ca = (GlobalBytes)(uintptr_t)codes + (uint32_t)stid;
                const uint32_t a_dd = (uint32_t)(de * rs) * 2u;
                uint32_t pw_addr = lds_addr(Pw), tw_addr = lds_addr(Tw);
                asm volatile("" : "+s"(pw_addr), "+s"(tw_addr));
                hot_cells(lo, width - 1, ca, lds_addr(p_oe) + ((uint32_t)rel_oe << 1), lds_addr(p_x) + ((uint32_t)rel_x << 1),
                          lds_addr(out_m) + ((uint32_t)rel_cur << 1), lds_addr(p_ip) + ((uint32_t)rel_e << 1), lds_addr(out_i) + ((uint32_t)rel_cur << 1),
                          lds_addr(p_ip) + a_dd + ((uint32_t)rel_e << 1), lds_addr(out_i) + a_dd + ((uint32_t)rel_cur << 1), pw_addr, tw_addr, 0u, touch);
              }
            } else {
              bool my_over = false;
              cells_of_score(std::false_type{}, lo, hi, codes, p_x + rel_x, p_oe + rel_oe - 1, p_ip + rel_e - 1, d_of(p_ip) + rel_e + 1,
                             out_m + rel_cur, out_i + rel_cur, out_d + rel_cur, my_over, touch,
                             e_x ? wx : ROW_NONE_A, e_oe ? wo : ROW_NONE_A, e_ie ? we : ROW_NONE_A);
            }
            bool any_touch;
            if constexpr (NW == 1) {
              block_sync<NW>();
              any_touch = touch != 0ull;
            } else {
              if (touch != 0ull && lane == 0) atomicOr(&bslot[1], 1u);
              __syncthreads();
              any_touch = bslot[1] != 0u;
            }
            touched_ever |= any_touch;
            // termination (:380-387): M[s][k*] == tlen
            if (touched_ever)
              done = ((unsigned)(kend - lo) <= (unsigned)(hi - lo)) && __builtin_amdgcn_readfirstlane((int)out_m[rel_cur + kend]) >= tlen;
            book.set_a(s & bkm, pack_range(lo, hi));
            if constexpr (NW == 1) block_sync<NW>();
            if (done) break;
          }
          break;
        }
        // ---- lean path: gap extension 1 and no cell has touched a sequence end yet.  Then no value can run past an end,
        // nothing is ever trimmed, and with e == 1 the limits of wavefront_compute.c:41-71 collapse to [lo - 1, hi + 1] of
        // the last score (the I and D rows of s-1 span its M limits), clipped by the window and the budget's reach.  It starts
        // at score 1: wavefronts and components that WFA2 does not create yet (no predecessor row exists) are computed here
        // as rows of NULL cells from the NULL rows of the ring -- a superset of WFA2's cells whose extra members are not
        // valid and only feed cells that are not valid either, at most one 64-lane chunk per early score.
        // An inner loop with its own small state: the instruction-issue pipes are what this kernel saturates, and
        // the scalar registers are what the compiler runs out of (every spilled one comes back through the vector unit).
        // ---- the same lean path written out for the one-wavefront 16-bit LDS tier (BASELINE's short-read configs live
        // here).  Vector and scalar instructions of a wavefront share its issue slots (one instruction per wave every
        // four cycles), so the per-score bookkeeping -- as many instructions as a 64-diagonal chunk of cells -- is cut
        // to the bone: limits in closed form, row addresses as one lane base (lo + lane - 1) plus a scalar slot base each
        // (every neighbour is a non-negative immediate away), row book written when the loop is left (its entries are
        // a function of the score), row-table entries by v_writelane, guard cells re-NULLed only once the budget's
        // reach makes the wavefront shrink (a growing wavefront overwrites everything its slot held before).
        if constexpr (HOT && !BANDED) {
          if (e == 1 && !touched_ever) {
            const int s_in = s, lo_in = last_lo, hi_in = last_hi;
            // lo(s) = max(lo_in - (s - s_in), wlo, s + c_lo), hi(s) = min(hi_in + (s - s_in), whi, c_hi - s)
            const int c_lo = bounded ? kend - budget : INT_MIN / 2, c_hi = bounded ? kend + budget : INT_MAX / 2;
            // first score at which a limit can move inwards (the reach bound has caught up with the window or the growing
            // front); from dm scores before... after it on, the slots written hold cells beyond the new limits
            int s_clear = INT_MAX;
            if (bounded) {
              const int t_lo = min(wlo - c_lo, (lo_in + s_in - c_lo) >> 1), t_hi = min(c_hi - whi, (c_hi - hi_in + s_in) >> 1);
              s_clear = min(t_lo, t_hi) - 1;
            }
            s_clear = __builtin_amdgcn_readfirstlane(s_clear);     // (a scalar, whatever unit the compiler formed it on)
            const uint32_t rsb = (uint32_t)rs * 2u;
            const uint32_t a_first = lds_addr(m_first), a_end = lds_addr(m_end);
            uint32_t a_m = lds_addr(p_m), a_x = lds_addr(p_x), a_oe = lds_addr(p_oe);
            uint32_t a_ic = lds_addr(p_ic), a_ip = lds_addr(p_ip);
            const uint32_t a_iswap = a_ic ^ a_ip;
            // the D rows of those I rows (hybrid ring: as byte offsets into the global D ring, which is laid out like the I ring)
            uint32_t a_dc = HYBRID ? a_ic - lds_addr(i_first) : a_ic + (uint32_t)(de * rs) * 2u,
                     a_dp = HYBRID ? a_ip - lds_addr(i_first) : a_ip + (uint32_t)(de * rs) * 2u;
            const uint32_t a_dswap = a_dc ^ a_dp;
            // (formed here, next to the loop that uses them in every chunk: defined further out they are the first
            // scalars the register allocator gives up, and every chunk then fetches them back from a vector register)
            uint32_t pw_addr = lds_addr(Pw), tw_addr = lds_addr(Tw), a_hm = lds_addr(hm_row0);
            asm volatile("" : "+s"(pw_addr), "+s"(tw_addr), "+s"(a_hm));
            int lo = lo_in, hi = hi_in;
            int t_lo = s_in + c_lo, t_hi = c_hi - s_in, f_lo = max(wlo, t_lo), f_hi = min(whi, t_hi);
            unsigned long long touch = 0;
            uint32_t a_last = a_m;
            // address of this lane's origin byte in the row of the current score (64-bit, bumped by the row size)
            GlobalBytes code_addr = nullptr;
            uint32_t need_prev = 0;
            if constexpr (BT) code_addr = (GlobalBytes)(uintptr_t)p.arena + ((size_t)chunk_cur * 16u + (uint32_t)stid);
            if constexpr (NW > 1) {
              // "a cell touched a sequence end": one LDS word, set by the waves that see it, read after the score's barrier
              if (tid == 0) bslot[1] = 0u;
              __syncthreads();
            }
            // why the loop ends: 1 = the reach interval is empty, 2 = arena exhausted, 3 = a cell touched a sequence end
            // (one exit at the bottom: several would be funnelled through a guard variable anyway)
            int why = 0;
            do {
              // (everything is updated in place -- no second set of registers to copy back at the bottom; when the loop
              // ends without having computed this score, the state of the last computed one is re-derived below)
              // (f_lo = max(wlo, s + c_lo) and f_hi = min(whi, c_hi - s), carried along: four loop constants fewer for the
              // scalar registers, which this loop runs out of -- a spilled one comes back through the vector unit)
              ++s; ++t_lo; --t_hi;
              f_lo = max(f_lo, t_lo); f_hi = min(f_hi, t_hi);
              asm volatile("" : "+s"(f_lo), "+s"(f_hi));   // (keeps the chains off v_max3/v_min3)
              lo = max(lo - 1, f_lo); hi = min(hi + 1, f_hi);
              if (__builtin_expect(lo > hi, 0)) { why = 1; continue; }
              const int wm1 = hi - lo;            // width - 1
              uint32_t need = 0;
              if constexpr (BT) {
                // the row of origin bytes (before anything of this score is committed: a failure leaves the loop right here)
                need = ((uint32_t)wm1 + 16u) >> 4;
                code_addr += need_prev << 4;
                if (__builtin_expect(need + WFA_ARENA_ROW_SLACK > chunk_left, 0)) {
                  const bool got = refill_arena(need + WFA_ARENA_ROW_SLACK);
                  code_addr = (GlobalBytes)(uintptr_t)cold_params()->arena + ((size_t)chunk_cur * 16u + (uint32_t)stid);
                  need_prev = 0;
                  if (!got) { why = 2; continue; }
                }
              }
              a_m += rsb;  if (a_m == a_end) a_m = a_first;
              a_x += rsb;  if (a_x == a_end) a_x = a_first;
              a_oe += rsb; if (a_oe == a_end) a_oe = a_first;
              a_ic ^= a_iswap; a_ip ^= a_iswap; a_dc ^= a_dswap; a_dp ^= a_dswap;
              ncells += (uint32_t)wm1;            // (+ 1 per score when the loop is left)
              if constexpr (BT) {
                row_s = chunk_cur; chunk_cur += need; chunk_left -= need; need_prev = need;
                if constexpr (NW == 1) {
                  // row table, buffered by lane: a new group of 64 scores starts at every multiple of 64
                  const int sl = s & 63;
                  if (sl == 0) tab[s - 64 + lane] = make_uint2((uint32_t)tabv_row, (uint32_t)tabv_lo);
                  asm("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                      : "+v"(tabv_row), "+v"(tabv_lo) : "s"((int)row_s), "s"(lo), "s"(sl));      // (m0: scratch, the compiler sets it before each use of its own)
                } else {
                  if (tid == 0) tab[s] = make_uint2(row_s, (uint32_t)lo);
                }
              }
              if (__builtin_expect(s >= s_clear, 0)) {
                // the slots written now last held scores s-dm (M) and s-2 (I, D), whose limits lay up to dm diagonals
                // further out: NULL the dm cells beyond each end (rows carry dm guard cells per side)
                // (lanes beyond 2 dm repeat the last of those cells: same value, same address, no exec mask)
                auto clear_guards = [&](const int j0) {
                  const int j = min(j0, 2 * dm - 1);
                  const int q = (j < dm) ? lo - 1 - j : hi + 1 - dm + j;
                  const uint32_t qa = (uint32_t)q << 1;
                  *(LdsRow)(qa + a_m) = (OffT)OffNull<OffT>::value;
                  *(LdsRow)(qa + a_ic) = (OffT)OffNull<OffT>::value;
                  if constexpr (HYBRID) *(GlobalRow)((GlobalBytes)(uintptr_t)d_first + (ptrdiff_t)(int32_t)(qa + a_dc)) = (OffT)OffNull<OffT>::value;
                  else *(LdsRow)(qa + a_dc) = (OffT)OffNull<OffT>::value;
                };
                clear_guards(tid);
                if constexpr (NW == 1) { if (__builtin_expect(2 * dm > 64, 0)) clear_guards(tid + 64); }      // (dm <= 64 in this tier)
              }
              hot_cells(lo, wm1, code_addr, a_oe, a_x, a_m, a_ip, a_ic, a_dp, a_dc, pw_addr, tw_addr, a_hm, touch);
              a_last = a_m;
              // a cell sits on a sequence end: it may be the last one (wavefront_extend.c:47-67), and from the next
              // score on values may run past the ends -- the careful path takes over
              if constexpr (NW == 1) {
                block_sync<NW>();
                if (touch != 0ull) why = 3;
              } else {
                if (touch != 0ull && lane == 0) atomicOr(&bslot[1], 1u);
                if constexpr (TIMED) {
                  const bool rec = blockIdx.x == 0 && lane == 0;
                  unsigned long long t_arrive = 0;
                  if (rec) t_arrive = __builtin_amdgcn_s_memtime();
                  __syncthreads();
                  if (rec) {
                    const unsigned long long t_leave = __builtin_amdgcn_s_memtime();
                    ColdParams cp = cold_params();
                    if (dbg_i < cp->dbg_cap) {
                      unsigned long long* rp = cp->dbg_times + ((size_t)dbg_i * NW + (size_t)(tid >> 6)) * 3;
                      rp[0] = t_arrive; rp[1] = t_leave; rp[2] = ((unsigned long long)(uint32_t)s << 32) | (uint32_t)wm1;
                    }
                    ++dbg_i;
                  }
                } else {
                  __syncthreads();
                }
                if (bslot[1] != 0u) why = 3;
              }
            } while (why == 0);
            if constexpr (NW > 1) {
              // the reduction slots of the careful loop (it resets the one of the next score as it goes)
              if (tid < 24) red[tid] = (tid & 7) == 6 ? 0 : (((tid & 7) & 1) ? INT_MIN : INT_MAX);
              __syncthreads();
            }
            const bool nomem = why == 2;
            if (why == 3) touched_ever = true;
            if (why != 3) {
              // the score at which the loop gave up was not computed: back to the limits of the one before
              --s;
              lo = max(max(lo_in - (s - s_in), wlo), s + c_lo); hi = min(min(hi_in + (s - s_in), whi), c_hi - s);
            }
            // back to the general state
            const int n_lean = s - s_in;
            ncells += (uint32_t)n_lean;
            if (touched_ever && n_lean > 0)
              done = ((unsigned)(kend - lo) <= (unsigned)(hi - lo)) &&
                     __builtin_amdgcn_readfirstlane((int)*(LdsRow)(a_last + ((uint32_t)kend << 1))) >= tlen;
            // row book: the last dm scores (older entries are never read again)
            for (int j = max(s_in + 1, s - dm + 1); j <= s; ++j) {
              const int jl = max(max(lo_in - (j - s_in), wlo), j + c_lo), jh = min(min(hi_in + (j - s_in), whi), c_hi - j);
              book.set_a(j & bkm, pack_range(jl, jh));
            }
            if (n_lean > 0) book.copy_a_to_id(tid, NT, bkm);
            regular += n_lean;
            last_lo = lo; last_hi = hi;
            p_m = m_first + (a_m - a_first) / 2; p_x = m_first + (a_x - a_first) / 2; p_oe = m_first + (a_oe - a_first) / 2;
            p_ic = m_first + (a_ic - a_first) / 2; p_ip = m_first + (a_ip - a_first) / 2;
            if (bounded) { rlo += n_lean; rhi -= n_lean; }
            tab_group = s >> 6;
            if (nomem) { status = WFA_ST_NOMEM; break; }
            if (done) break;
          }
        }
        // ---- lean path, any gap extension: the same cells, but the limits come from the row book (three reads:
        // lo = min(lo[s-x], lo[s-o-e] - 1, lo[s-e] - 1), hi alike -- every row that exists carries all three components over
        // its limits here), the reach interval moves one diagonal every e scores, and scores without any predecessor row
        // (all odd scores of an all-even penalty set) are "no wavefront" scores: their ring slots are cleared and nothing
        // is computed.
        if constexpr (HOT && !BANDED) {
          // (the same trimmed bookkeeping as the e == 1 loop above, except that the limits come from the row book)
          if (e != 1 && !touched_ever) {
            const int s_in = s;
            const uint32_t rsb = (uint32_t)rs * 2u;
            const uint32_t a_first = lds_addr(m_first), a_end = lds_addr(m_end), ai_first = lds_addr(i_first), ai_end = lds_addr(i_end);
            uint32_t a_m = lds_addr(p_m), a_x = lds_addr(p_x), a_oe = lds_addr(p_oe), a_ic = lds_addr(p_ic), a_ip = lds_addr(p_ip);
            // the D row of an I row: further up in LDS, or (hybrid ring) the same offset into the global D ring
            const uint32_t d_delta = HYBRID ? 0u - ai_first : (uint32_t)(de * rs) * 2u;
            uint32_t pw_addr = lds_addr(Pw), tw_addr = lds_addr(Tw), a_hm = lds_addr(hm_row0);
            asm volatile("" : "+s"(pw_addr), "+s"(tw_addr), "+s"(a_hm));
            unsigned long long touch = 0;
            uint32_t a_last = a_m;
            GlobalBytes code_addr = nullptr;
            uint32_t need_prev = 0;
            if constexpr (BT) code_addr = (GlobalBytes)(uintptr_t)p.arena + ((size_t)chunk_cur * 16u + (uint32_t)stid);
            if constexpr (NW > 1) {
              if (tid == 0) bslot[1] = 0u;
              __syncthreads();
            }
            int lo = 0, hi = -1;
            int prev_lo = last_lo;        // (banded: lower limit of the last wavefront that exists)
            uint32_t o_cur = 0;           // (banded: byte offset of diagonal 0 in the row written last: (GZ - lo) * 2)
            // why the loop ends: 1 = budget exhausted, 2 = arena exhausted, 3 = a cell touched a sequence end
            int why = 0;
            do {
              const int ns = s + 1;
              int n_rr = reach_r, n_rlo = rlo, n_rhi = rhi;
              if (n_rr == 0) { ++n_rlo; --n_rhi; n_rr = e - 1; } else --n_rr;
              const int b_x = book.get_a((ns - x) & bkm), b_oe = book.get_a((ns - oe) & bkm), b_e = book.get_a((ns - e) & bkm);
              lo = min(range_lo(b_x), range_lo(b_oe) - 1); hi = max(range_hi(b_x), range_hi(b_oe) + 1);
              if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
              lo = min(lo, range_lo(b_e) - 1); hi = max(hi, range_hi(b_e) + 1);
              if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
              lo = max(lo, wlo); hi = min(hi, whi);
              if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
              lo = max(lo, n_rlo); hi = min(hi, n_rhi);
              const bool none = lo > hi;
              if (__builtin_expect(none && bounded && ns > budget, 0)) { why = 1; continue; }      // the careful path reports it
              const int wm1 = hi - lo;
              uint32_t need = 0;
              if constexpr (BT) {
                if (!none) {
                  need = ((uint32_t)wm1 + 16u) >> 4;
                  code_addr += need_prev << 4;
                  if (__builtin_expect(need + WFA_ARENA_ROW_SLACK > chunk_left, 0)) {
                    const bool got = refill_arena(need + WFA_ARENA_ROW_SLACK);
                    code_addr = (GlobalBytes)(uintptr_t)cold_params()->arena + ((size_t)chunk_cur * 16u + (uint32_t)stid);
                    need_prev = 0;
                    if (!got) { why = 2; continue; }
                  }
                }
              }
              s = ns; rlo = n_rlo; rhi = n_rhi; reach_r = n_rr;
              a_m += rsb;  if (a_m == a_end) a_m = a_first;
              a_x += rsb;  if (a_x == a_end) a_x = a_first;
              a_oe += rsb; if (a_oe == a_end) a_oe = a_first;
              a_ic += rsb; if (a_ic == ai_end) a_ic = ai_first;
              a_ip += rsb; if (a_ip == ai_end) a_ip = ai_first;
              const uint32_t a_dc = a_ic + d_delta, a_dp = a_ip + d_delta;
              // (banded: rows relative to their own lower limit -- the per-row offsets of diagonal 0; a row that does not
              // exist reads through the mapping of the current one: its slot is NULL all over)
              uint32_t o_x = 0, o_oe = 0, o_e = 0;
              if constexpr (BANDED) {
                o_cur = (uint32_t)(GZ - lo) << 1;
                o_x = range_lo(b_x) <= range_hi(b_x) ? (uint32_t)(GZ - range_lo(b_x)) << 1 : o_cur;
                o_oe = range_lo(b_oe) <= range_hi(b_oe) ? (uint32_t)(GZ - range_lo(b_oe)) << 1 : o_cur;
                o_e = range_lo(b_e) <= range_hi(b_e) ? (uint32_t)(GZ - range_lo(b_e)) << 1 : o_cur;
              }
              auto store_null = [&](const uint32_t qa) {
                *(LdsRow)(qa + a_m) = (OffT)OffNull<OffT>::value;
                *(LdsRow)(qa + a_ic) = (OffT)OffNull<OffT>::value;
                if constexpr (HYBRID) *(GlobalRow)((GlobalBytes)(uintptr_t)d_first + (ptrdiff_t)(int32_t)(qa + a_dc)) = (OffT)OffNull<OffT>::value;
                else *(LdsRow)(qa + a_dc) = (OffT)OffNull<OffT>::value;
              };
              if (none) {
                // no wavefront at this score: the slots it would have written must read as NULL everywhere
                if constexpr (BANDED) {
                  for (int q = tid; q < p.band_width + 2 * GZ; q += NT) store_null((uint32_t)q << 1);      // (the whole slot)
                } else {
                  const int o_m = book.get_a((s - dm) & bkm), o_e = book.get_a((s - de) & bkm);
                  const int f0 = min(range_lo(o_m), range_lo(o_e)), f1 = max(range_hi(o_m), range_hi(o_e));
                  for (int q = f0 + tid; q <= f1; q += NT) store_null((uint32_t)q << 1);
                }
                book.set_a(s & bkm, ROW_NONE_A);
                block_sync<NW>();
                continue;
              }
              ncells += (uint32_t)wm1 + 1u;
              if constexpr (BT) {
                row_s = chunk_cur; chunk_cur += need; chunk_left -= need; need_prev = need;
                tab_set(s, row_s, lo);
              }
              if constexpr (BANDED) {
                // ring invariant of relative rows: the guard zones on both sides of the row (columns 0 .. GZ-1 and GZ beyond
                // its last cell) read NULL whatever the slot held before; nothing further out is ever read (see GZ)
                for (int j = tid; j < 2 * GZ; j += NT) store_null((uint32_t)(j < GZ ? j : wm1 + 1 + j) << 1);
                prev_lo = lo;
              } else {
                // ring invariant: the limits move by at most one diagonal per score (see the e == 1 loop): dm cells beyond each end
                // (lanes beyond 2 dm repeat the last of those cells: same value, same address, no exec mask)
                auto clear_guards = [&](const int j0) {
                  const int j = min(j0, 2 * dm - 1);
                  store_null((uint32_t)((j < dm) ? lo - 1 - j : hi + 1 - dm + j) << 1);
                };
                clear_guards(tid);
                if constexpr (NW == 1) { if (__builtin_expect(2 * dm > 64, 0)) clear_guards(tid + 64); }
              }
              hot_cells(lo, wm1, code_addr, a_oe + o_oe, a_x + o_x, a_m + o_cur, a_ip + o_e, a_ic + o_cur, a_dp + o_e, a_dc + o_cur, pw_addr, tw_addr, a_hm, touch);
              a_last = a_m + o_cur;
              if constexpr (NW == 1) {
                block_sync<NW>();
                if (touch != 0ull) why = 3;
              } else {
                if (touch != 0ull && lane == 0) atomicOr(&bslot[1], 1u);
                __syncthreads();
                if (bslot[1] != 0u) why = 3;
              }
              book.set_a(s & bkm, pack_range(lo, hi));
              if constexpr (NW == 1) block_sync<NW>();
            } while (why == 0);
            if constexpr (NW > 1) {
              // the reduction slots of the careful loop (it resets the one of the next score as it goes)
              if (tid < 24) red[tid] = (tid & 7) == 6 ? 0 : (((tid & 7) & 1) ? INT_MIN : INT_MAX);
              __syncthreads();
            }
            if (why == 3) {
              // a cell sits on a sequence end: it may be the last one (wavefront_extend.c:47-67), and from the next
              // score on values may run past the ends -- the careful path takes over
              touched_ever = true;
              done = ((unsigned)(kend - lo) <= (unsigned)(hi - lo)) &&
                     __builtin_amdgcn_readfirstlane((int)*(LdsRow)(a_last + ((uint32_t)kend << 1))) >= tlen;
            }
            if (s != s_in) book.copy_a_to_id(tid, NT, bkm);
            if constexpr (BANDED) last_lo = prev_lo;
            regular = 0;      // (the careful path's shortcut for runs of regular scores starts counting afresh)
            p_m = m_first + (a_m - a_first) / 2; p_x = m_first + (a_x - a_first) / 2; p_oe = m_first + (a_oe - a_first) / 2;
            p_ic = m_first + (a_ic - a_first) / 2; p_ip = m_first + (a_ip - a_first) / 2;
            if (why == 2) { status = WFA_ST_NOMEM; break; }
            if (done) break;
          }
        }
        if constexpr (!HOT) {
          // (the tiers whose ring lives in HBM: one loop for every gap extension -- they are bound by the ring traffic, a
          // closed-form e == 1 twin of it bought nothing there)
          if (!touched_ever) {
            const int s_in = s;
            bool nomem = false;
            for (;;) {
              const int ns = s + 1;
              int n_rr = reach_r, n_rlo = rlo, n_rhi = rhi;
              if (n_rr == 0) { ++n_rlo; --n_rhi; n_rr = e - 1; } else --n_rr;
              const int a_x = book.get_a((ns - x) & bkm), a_oe = book.get_a((ns - oe) & bkm), a_e = book.get_a((ns - e) & bkm);
              int lo = min(range_lo(a_x), range_lo(a_oe) - 1), hi = max(range_hi(a_x), range_hi(a_oe) + 1);
              if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
              lo = min(lo, range_lo(a_e) - 1); hi = max(hi, range_hi(a_e) + 1);
              if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
              lo = max(lo, wlo); hi = min(hi, whi);
              if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
              lo = max(lo, n_rlo); hi = min(hi, n_rhi);
              if (lo > hi && bounded && ns > budget) break;      // budget exhausted: the careful path reports it
              s = ns; rlo = n_rlo; rhi = n_rhi; reach_r = n_rr;
              if constexpr (NW > 1) {
                if (tid < 8) red[8 * ((s + 1) % 3) + tid] = (tid == 6) ? 0 : ((tid & 1) ? INT_MIN : INT_MAX);
              }
              p_m += rs;  if (p_m == m_end) p_m = m_first;
              p_x += rs;  if (p_x == m_end) p_x = m_first;
              p_oe += rs; if (p_oe == m_end) p_oe = m_first;
              p_ic += rs; if (p_ic == i_end) p_ic = i_first;
              p_ip += rs; if (p_ip == i_end) p_ip = i_first;
              OffT* out_m = p_m; OffT* out_i = p_ic; OffT* out_d = d_of(p_ic);
              if (lo > hi) {
                // no wavefront at this score: the slots it would have written must read as NULL everywhere
                const int o_m = book.get_a((s - dm) & bkm), o_e = book.get_a((s - de) & bkm);
                const int f0 = min(range_lo(o_m), range_lo(o_e)), f1 = max(range_hi(o_m), range_hi(o_e));
                for (int q = f0 + tid; q <= f1; q += NT) {
                  out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
                }
                book.set_a(s & bkm, ROW_NONE_A);
                block_sync<NW>();
                continue;
              }
              const int width = hi - lo + 1;
              ncells += (uint32_t)width;
              uint8_t* codes = nullptr;
              if constexpr (BT) {
                if (!alloc_row(width)) { nomem = true; break; }
                tab_set(s, row_s, lo);
                codes = p.arena + (size_t)row_s * 16;
              }
              // ring invariant: the limits move by at most one diagonal per score (see the e == 1 loop)
              for (int j0 = 0; j0 < 2 * dm; j0 += NT) {
                const int j = min(j0 + tid, 2 * dm - 1);
                const int q = (j < dm) ? lo - 1 - j : hi + 1 - dm + j;
                out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
              }
              bool my_over = false;
              unsigned long long touch_mask = 0;
              cells_of_score(std::true_type{}, lo, hi, codes, p_x, p_oe - 1, p_ip - 1, d_of(p_ip) + 1, out_m, out_i, out_d,
                             my_over, touch_mask);
              const bool wave_touch = touch_mask != 0ull;
              bool any_touch;
              if constexpr (NW == 1) {
                block_sync<NW>();
                any_touch = wave_touch;
              } else {
                int* acc = red + 8 * (s % 3);
                if (lane == 0 && wave_touch) atomicOr(&acc[6], 4);
                __syncthreads();
                any_touch = (acc[6] & 4) != 0;
              }
              book.set_a(s & bkm, pack_range(lo, hi));
              if constexpr (NW == 1) block_sync<NW>();
              if (any_touch) {
                touched_ever = true;
                done = ((unsigned)(kend - lo) <= (unsigned)(hi - lo)) && __builtin_amdgcn_readfirstlane((int)out_m[kend]) >= tlen;
                break;
              }
            }
            if (s != s_in) book.copy_a_to_id(tid, NT, bkm);
            regular = 0;      // (the careful path's shortcut for runs of regular scores starts counting afresh)
            if (nomem) { status = WFA_ST_NOMEM; break; }
            if (done) break;
          }
        }
        ++s;
        // (past the budget the reach interval is empty, so that test sits on the "no wavefront" path)
        if (reach_r == 0) { ++rlo; --rhi; reach_r = e - 1; } else --reach_r;
        if constexpr (NW > 1) {
          // reduction slot of the NEXT score (nobody reads it any more: its readers passed barrier s-1)
          if (tid < 8) red[8 * ((s + 1) % 3) + tid] = (tid == 6) ? 0 : ((tid & 1) ? INT_MIN : INT_MAX);
        }
        p_m += rs;  if (p_m == m_end) p_m = m_first;
        p_x += rs;  if (p_x == m_end) p_x = m_first;
        p_oe += rs; if (p_oe == m_end) p_oe = m_first;
        p_ic += rs; if (p_ic == i_end) p_ic = i_first;
        p_ip += rs; if (p_ip == i_end) p_ip = i_first;
        const int bk_s = s & bkm, bk_x = (s - x) & bkm, bk_oe = (s - oe) & bkm, bk_e = (s - e) & bkm;
        // predecessor rows: s-x and s-(o+e) of M, s-e of I and D
        const int a_x = book.get_a(bk_x);
        const int a_oe = book.get_a(bk_oe);
        const int mxlo = range_lo(a_x), mxhi = range_hi(a_x), molo = range_lo(a_oe), mohi = range_hi(a_oe);
        // limits (wavefront_compute.c:41-71; null rows carry lo=1, hi=-1)
        // (the empty asm keeps the chains on the scalar unit: min(min(a,b),c) would be matched to v_min3)
        int lo = min(mxlo, molo - 1), hi = max(mxhi, mohi + 1);
        if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
        bool mx_null = false, mo_null = false, ie_null = false, de_null = false;
        bool all_null = false, have_i = true, have_d = true;
        int ielo, iehi, delo, dehi;
        const bool fast = regular >= dm - 1;
        if (fast) {
          // I and D of s-e span the M limits of s-e: min(lo+1, lo-1), max(hi+1, hi-1)
          const int a_e = book.get_a(bk_e);
          ielo = delo = range_lo(a_e); iehi = dehi = range_hi(a_e);
          lo = min(lo, ielo - 1); hi = max(hi, iehi + 1);
        } else {
          const int bi_e = book.get_i(bk_e);
          const int bd_e = book.get_d(bk_e);
          ielo = range_lo(bi_e); iehi = range_hi(bi_e); delo = range_lo(bd_e); dehi = range_hi(bd_e);
          mx_null = mxlo > mxhi; mo_null = molo > mohi; ie_null = ielo > iehi; de_null = delo > dehi;
          all_null = mx_null && mo_null && ie_null && de_null;
          have_i = !(mo_null && ie_null); have_d = !(mo_null && de_null);
          lo = min(lo, ielo + 1); hi = max(hi, iehi + 1);
          if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
          lo = min(lo, delo - 1); hi = max(hi, dehi - 1);
        }
        {
          if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
          lo = max(lo, wlo); hi = min(hi, whi);
          if constexpr (NW == 1) asm volatile("" : "+s"(lo), "+s"(hi));
          lo = max(lo, rlo); hi = min(hi, rhi);
        }
        OffT* out_m = p_m;             // exact mode: [q] = diagonal q
        OffT* out_i = p_ic;
        OffT* out_d = d_of(p_ic);
        if (all_null || lo > hi) {
          // no wavefront at this score (wavefront_compute_affine.c:236-243)
          if (s > budget) { status = WFA_ST_SCORE; break; }
          regular = 0;
          book.set(bk_s, ROW_NONE_A, ROW_NONE_A, ROW_NONE_A);
          if constexpr (BANDED) {
            // (relative rows: the whole slot)
            for (int q = tid; q < p.band_width + 2 * GZ; q += NT) {
              out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
            }
          } else {
            // the slots this score would have written: clear what their previous occupants left
            const int o_m = book.get_a((s - dm) & bkm), o_e = book.get_a((s - de) & bkm);
            const int f0 = min(range_lo(o_m), range_lo(o_e)), f1 = max(range_hi(o_m), range_hi(o_e));
            for (int q = f0 + tid; q <= f1; q += NT) {
              out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
            }
          }
          block_sync<NW>();
          continue;
        }
        // (the banded search has a loop of its own above and never gets here)
        const int width = hi - lo + 1;
        ncells += (uint32_t)width;

        uint8_t* codes = nullptr;
        if constexpr (BT) {
          if (!alloc_row(width)) { status = WFA_ST_NOMEM; break; }
          tab_set(s, row_s, lo);
          codes = p.arena + (size_t)row_s * 16;
        }

        const OffT* row_mx = p_x;
        const OffT* row_mo = p_oe;
        const OffT* row_ie = p_ip;
        const OffT* row_de = d_of(p_ip);

        if constexpr (BANDED) {
          // ring invariant of relative rows: NULL guard zones on both sides of the row, whatever the slot held before
          for (int j = tid; j < 2 * GZ; j += NT) {
            const int q = j < GZ ? j : width + j;
            out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
          }
        } else {
          // Keep the ring invariant: the M slot last held score s-dm, the I/D slots score s-e-1; whatever
          // those rows had beyond [lo, hi] becomes NULL again (nothing while the wavefront grows).
          const int o_m = book.get_a((s - dm) & bkm), o_e = book.get_a((s - de) & bkm);
          const int f0 = min(range_lo(o_m), range_lo(o_e)), f1 = max(range_hi(o_m), range_hi(o_e));
          const int nlo = max(lo - f0, 0), ntot = nlo + max(f1 - hi, 0);
          if (ntot > 0) {
            for (int j = tid; j < ntot; j += NT) {
              const int q = (j < nlo) ? f0 + j : hi + 1 + (j - nlo);
              out_m[q] = (OffT)OffNull<OffT>::value; out_i[q] = (OffT)OffNull<OffT>::value; out_d[q] = (OffT)OffNull<OffT>::value;
            }
          }
        }

        // ---- the cells of this score.  Lanes past the end recompute cell `hi` (same values, same
        // addresses), so no store needs an exec mask; the only inner loop (extend) is wave-uniform.
        // The vector ALU is the unit this kernel saturates (one integer wave64 instruction holds its
        // SIMD for 4 cycles), so everything uniform is folded into scalar row bases: each LDS address
        // is one v_lshl_add of the diagonal.
        // (banded: a row is stored relative to its own lower limit -- diagonal 0 of a row with limits [lo_r, ..] is column
        // GZ - lo_r of its slot; M, I and D of a score share the mapping; a row that does not exist is read through the
        // mapping of the current one: its slot is NULL all over)
        const int rel_cur = BANDED ? GZ - lo : 0;
        const int rel_x = BANDED ? (mxlo <= mxhi ? GZ - mxlo : rel_cur) : 0;
        const int rel_oe = BANDED ? (molo <= mohi ? GZ - molo : rel_cur) : 0;
        int rel_e = 0;
        if constexpr (BANDED) { const int a_e2 = book.get_a(bk_e); rel_e = range_lo(a_e2) <= range_hi(a_e2) ? GZ - range_lo(a_e2) : rel_cur; }
        const OffT* rb_mx = row_mx + rel_x;          // [k]
        const OffT* rb_mo = row_mo + rel_oe - 1;      // [k] = k-1, [k+2] = k+1
        const OffT* rb_ie = row_ie + rel_e - 1;      // [k] = k-1
        const OffT* rb_de = row_de + rel_e + 1;      // [k] = k+1
        OffT* wb_m = out_m + rel_cur;
        OffT* wb_i = out_i + rel_cur;
        OffT* wb_d = out_d + rel_cur;
        bool my_over = false;
        unsigned long long touch_mask = 0;
        cells_of_score(std::false_type{}, lo, hi, codes, rb_mx, rb_mo, rb_ie, rb_de, wb_m, wb_i, wb_d, my_over, touch_mask);
        bool any_over = false;
        {
          const bool wave_over = __builtin_amdgcn_ballot_w64(my_over) != 0ull;
          const bool wave_touch = touch_mask != 0ull;
          if constexpr (NW == 1) {
            block_sync<NW>();
            any_over = wave_over;
            touched_ever |= wave_touch;
          } else {
            int* acc = red + 8 * (s % 3);
            if (lane == 0 && (wave_over || wave_touch)) atomicOr(&acc[6], (wave_over ? 2 : 0) | (wave_touch ? 4 : 0));
            __syncthreads();
            any_over = (acc[6] & 2) != 0;
            touched_ever |= (acc[6] & 4) != 0;
          }
          // termination (wavefront_extend.c:47-67): every lane reads the same cell
          done = ((unsigned)(kend - lo) <= (unsigned)(hi - lo)) && __builtin_amdgcn_readfirstlane((int)wb_m[kend]) >= tlen;
        }
        // Limits recorded for the row: the computed ones.  Cells that are not valid hold NULL or a
        // negative value, which is all a reader needs; only values past a sequence end need the
        // exact per-component trimming (wavefront_compute.c:570-603): first/last in-range cell.
        if (fast && !any_over) {
          // regular regime goes on: the three components exist and span the computed limits
          const int lim = pack_range(lo, hi);
          ++regular;
          book.set(bk_s, lim, lim, lim);
          last_lo = lo; last_hi = hi;
          if constexpr (NW == 1) block_sync<NW>();
          if (done) break;
          continue;
        }
        int lim_i = have_i ? pack_range(lo, hi) : ROW_NONE_A, lim_d = have_d ? pack_range(lo, hi) : ROW_NONE_A;
        if (any_over) {
          const int wave = (NW == 1) ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
          int r[4] = {INT_MAX, INT_MIN, INT_MAX, INT_MIN};
          for (int k0 = lo; k0 <= hi; k0 += NT) {
            const int kraw = k0 + tid;
            const bool active = kraw <= hi;
            const int k = active ? kraw : hi;
            const int iv = have_i ? (int)wb_i[k] : OffNull<OffT>::value;
            const int dv = have_d ? (int)wb_d[k] : OffNull<OffT>::value;
            const bool i_ok = ((unsigned)iv <= (unsigned)tlen) && ((unsigned)(iv - k) <= (unsigned)plen);
            const bool d_ok = ((unsigned)dv <= (unsigned)tlen) && ((unsigned)(dv - k) <= (unsigned)plen);
            const int b = k0 + wave * 64;
            const unsigned long long bi = __ballot(active && i_ok), bd = __ballot(active && d_ok);
            if (bi) { r[0] = min(r[0], b + (int)__builtin_ctzll(bi)); r[1] = max(r[1], b + 63 - (int)__builtin_clzll(bi)); }
            if (bd) { r[2] = min(r[2], b + (int)__builtin_ctzll(bd)); r[3] = max(r[3], b + 63 - (int)__builtin_clzll(bd)); }
          }
          if constexpr (NW > 1) {
            int* acc = red + 8 * (s % 3) + 2;            // words 2..5 of this score's slot are unused so far
            if (lane == 0) {
              if (r[0] <= r[1]) { atomicMin(&acc[0], r[0]); atomicMax(&acc[1], r[1]); }
              if (r[2] <= r[3]) { atomicMin(&acc[2], r[2]); atomicMax(&acc[3], r[3]); }
            }
            __syncthreads();
            r[0] = acc[0]; r[1] = acc[1]; r[2] = acc[2]; r[3] = acc[3];
          }
          if (r[0] > r[1]) { r[0] = ROW_NONE_LO; r[1] = ROW_NONE_HI; }
          if (r[2] > r[3]) { r[2] = ROW_NONE_LO; r[3] = ROW_NONE_HI; }
          lim_i = pack_range(r[0], r[1]); lim_d = pack_range(r[2], r[3]);
          {
            // trimmed-away cells read as NULL from now on (wavefront_compute.c:480-520)
            for (int q = lo + tid; q <= hi; q += NT) {
              if (q < r[0] || q > r[1]) wb_i[q] = (OffT)OffNull<OffT>::value;
              if (q < r[2] || q > r[3]) wb_d[q] = (OffT)OffNull<OffT>::value;
            }
            if constexpr (NW > 1) __syncthreads();
          }
        }
        regular = (have_i && have_d && !any_over) ? regular + 1 : 0;
        book.set(bk_s, pack_range(lo, hi), lim_i, lim_d);
        last_lo = lo; last_hi = hi;
        if constexpr (NW == 1) block_sync<NW>();
        if (done) break;
      }
      if constexpr (BT) {
        if (status == WFA_ST_DONE) {
          tab_flush(s);
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
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), NW * 64, lds) != hipSuccess) nb = 0;
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
template <bool BT>
void launch_tier_banded(const WfaAlignParams& p, int tier, size_t lds, int grid, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  switch (tier) {
    case 0: launch_inst<1, BT, int16_t, false, false, true>(p, lds, grid, stream, ev0, ev1); break;
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
    case 4: if constexpr (!RAW) return occ_inst<16, BT, int16_t, false, false, false, true>(lds); else return 0;
    default: return occ_inst<16, BT, int32_t, true, RAW, false>(lds);
  }
}
template <bool BT>
int occ_tier_banded(int tier, size_t lds) {
  switch (tier) {
    case 0: return occ_inst<1, BT, int16_t, false, false, true>(lds);
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
  const size_t meta = (size_t)(24 + 2 + (tier == 0 ? 0 : 3 * bk)) * 4;     // (tiers 1, 2, 3, 4: row book in LDS)
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
