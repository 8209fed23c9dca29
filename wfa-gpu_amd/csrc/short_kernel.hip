// Gap-affine WFA for SHORT wavefronts: several alignments per 64-lane wavefront, rings in registers (tier 5).
//
// The one-wave tier of align_kernel.hip gives every alignment a whole wavefront: at BASELINE configs[1] (150 bp reads at
// 2 % error: optimal scores of 2..12, a score budget of ~14 once it is tuned) a wavefront row is 5..13 diagonals wide, so
// 80 % of the lanes carry nothing and the per-score bookkeeping -- as many instructions as a 64-diagonal chunk of cells --
// is paid per alignment.  Here a group of L = 16 (or 32) lanes is one alignment, lane j of the group is diagonal wlo + j for
// the WHOLE alignment (the exact diagonal window of the pair's score budget must fit the group: the same window argument as
// in align_kernel.hip), and the wavefront history a cell needs -- M of the last D = max(x, o+e) <= 8 scores, I and D of the
// last one (e == 1) or of the last D (e = 2..4, round 6: short_kernel_e2/3/4.hip) -- lives in registers of the lane itself:
// M[s-x][k] is a register, M[s-o-e][k-1], I[s-e][k-1], M[s-o-e][k+1], D[s-e][k+1] are DPP row shifts (v_mov_b32_dpp row_shr:1 / row_shl:1: a "row" of the DPP network is 16
// lanes = one group; the lanes a shift cannot feed keep NULL) of the neighbours' registers.  No LDS ring, no row limits, no
// ring invariant, no barrier; LDS only holds the packed sequences of the G = 64 / L pairs for the extend.  The score loop is
// unrolled over the ring depth so that every register index is a compile-time constant.
//
// Replaces, for such batches, the reference's distance_kernel (lib/kernels/sequence_distance_kernel.cu:175-425) and, with
// CIGARs (BT), its alignment_kernel (sequence_alignment_kernel.cu:355-688).  Results: the optimal gap-affine score
// (wavefront_compute_affine.c:45-87 recurrences, out-of-range values are NULL, termination M[s][tlen - plen] >= tlen:
// wavefront_extend.c:47-67) and WFA2's CIGAR (see the note at the kernel).  Pairs whose window does not fit a group
// (status BAND), whose score exceeds the budget (status SCORE) or that find the arena full (status NOMEM) go on to the
// ordinary tiers / the next pass on the device like the failures of any tier.
#include "short_kernel_impl.h"

namespace {
const WfaShortEntry& short_pick(const WfaAlignParams& p, int lanes, bool with_bt) {
  const int idx = (p.x - 1) * 8 + (p.oe - 1), a = p.ascii ? 1 : 0, b = with_bt ? 1 : 0, l = lanes == 16 ? 0 : 1;
  switch (p.e) {
    case 1: return *ShortTables<1>::pick(a, b, l, idx);
    case 2: return *wfa_short_entry_e2(a, b, l, idx);
    case 3: return *wfa_short_entry_e3(a, b, l, idx);
    default: return *wfa_short_entry_e4(a, b, l, idx);
  }
}
}  // namespace

// arena units (16 bytes) a work item of a CIGAR launch owns: row table up to max_score + its rows
unsigned long long wfa_short_bt_slot_units(int max_score, int lanes) {
  return (unsigned long long)(((long long)max_score + 2) >> 1) + (unsigned long long)(max_score + 1) * (unsigned)(lanes / 16);
}

// (after the common-factor reduction: gap extensions to 4, history of eight scores)
bool wfa_short_supported(int x, int oe, int e) { return e >= 1 && e <= 4 && oe >= e && x >= 1 && x <= 8 && oe <= 8; }

// the staged sequences of the 64 / lanes pairs of a wavefront; with_bt: + their rows of origin bytes up to the launch's largest budget
size_t wfa_short_lds_bytes(const WfaAlignParams& p, int lanes, bool with_bt) {
  const size_t seq = 2 * ((((size_t)(64 / lanes) * 2 * p.seq_words_cap) + 3) & ~(size_t)3) * 4;      // (two sets: this iteration's and the next one's)
  return seq + (with_bt ? (size_t)64 * (size_t)(p.max_score + 1) : 0);
}

void wfa_launch_short(const WfaAlignParams& p, int lanes, bool with_bt, int grid, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  short_pick(p, lanes, with_bt).launch(p, grid, stream, ev0, ev1);
}

int wfa_short_max_blocks_per_cu(const WfaAlignParams& p, int lanes, bool with_bt) { return short_pick(p, lanes, with_bt).occ(wfa_short_lds_bytes(p, lanes, with_bt)); }

// Loads this translation unit's code object on the current device (the runtime loads a code object at the first launch of
// any of its kernels: 5-25 ms each): launch_alignments* call it while a cold call waits for its first upload.
namespace { __global__ void k_prime_short() {} }
void wfa_prime_short(hipStream_t stream) { hipLaunchKernelGGL(k_prime_short, dim3(1), dim3(64), 0, stream); }
