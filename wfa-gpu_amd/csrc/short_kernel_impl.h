// short_kernel_impl.h -- the kernel of tier 5 (several alignments per wavefront, rings in registers) and its tables, for ONE gap extension E.
// Included by short_kernel.hip (E = 1: BASELINE configs[1], and everything around the kernels) and by short_kernel_e2/3/4.hip: a
// translation unit -- a code object, compiled in parallel, loaded at its first launch -- per gap extension.
#pragma once
#include <array>
#include <utility>

#include "wfa_device.h"
#include "pack_device.h"

namespace {

constexpr int S_NULL = -(1 << 28);

// value of the lane one diagonal below (k - 1) / above (k + 1) within the group; lanes without such a neighbour get NULL
template <int L> __device__ __forceinline__ int from_below(int v, int j) {
  if constexpr (L == 16) return __builtin_amdgcn_update_dpp(S_NULL, v, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
  else { const int r = __shfl_up(v, 1, L); return j == 0 ? S_NULL : r; }
}
template <int L> __device__ __forceinline__ int from_above(int v, int j) {
  if constexpr (L == 16) return __builtin_amdgcn_update_dpp(S_NULL, v, 0x101 /* row_shl:1 */, 0xF, 0xF, false);
  else { const int r = __shfl_down(v, 1, L); return j == L - 1 ? S_NULL : r; }
}

// X = mismatch, OE = gap open + extend, E = gap extend.  L lanes per alignment.  The history a cell reads -- M of the last
// D = max(X, OE) scores -- is a ring of D registers per lane; the score loop is unrolled D times so that every ring index is a
// compile-time constant (round 3 had the two instantiations of the benchmark penalties; round 4 every set with e == 1 and
// max(x, o + e) <= 8 after the common-factor reduction; round 6 gap extensions to 4: 26 (o + e, e) x 8 x sets, picked from tables,
// one translation unit per gap extension).
// BT (round 4): with CIGARs.  Every cell also leaves its origin byte (wfa_device.h: source of M with WFA2's priority on equal
// offsets -- mismatch, then deletion, then insertion --, gap extension over gap open) in a row of L bytes per score; the rows
// collect in LDS and go out, with the row table, when the alignment is done, into a slot of the arena the work item owns
// without any atomic (a few hundred bytes: budgets are <= 35 here); the backtrace kernels (trace_kernel.hip) read them like
// any other tier's.
// Is it WFA2's CIGAR?  This kernel has no per-row limits and turns every I / D value that ran past a sequence end into NULL
// at once, where WFA2 keeps such values inside a row and trims them at its ends only (wavefront_compute.c:570-603).  The
// two differ only in cells that no optimal alignment passes through (a value past an end cannot lie on a path to the corner,
// and every candidate of a valid cell is itself valid or NULL: a nulled value never was the winning candidate of a cell that
// stays valid in WFA2), so scores, tie-breaks along the optimal path and hence the CIGAR are the same: argued in DESIGN.md
// section 4.2b, searched with scratch/short_cigar_semantics.py (oracle with that one change: 0 of 240 000 indel-heavy short
// pairs differ), and held by the parity tests, which compare every CIGAR of this tier with WFA2's.
// ASCII: the batch has not been packed (WfaAlignParams::ascii): the sixteen bytes of a word are what is requested ahead, and the word
// is made when it is staged -- to LDS and to the packed buffer, where the backtrace kernels look for it.  (A pack kernel in front of a
// launch of 100k configs[1] pairs was 18 us + the gap between two launches of a 168 us step; here it is ~80 vector instructions per
// iteration in a kernel whose vector pipe is half idle.)  A pair with a byte outside ACGT leaves with status ALPHABET like in the
// wavefront kernels that pack while staging (align_kernel.hip).
// (register budget: the CIGAR variants came out one and three registers above a step of the occupancy table -- 81 and 99 for x, o, e =
// 2, 3, 1 -- and are compiled for the step below: 1M configs[1] pairs with CIGARs 1.096 -> 1.045 ms per step, 100k the same; the
// score-only variants sit on their steps, 72 and 80, by themselves)
// E > 1 (round 6; the reference's kernels take any penalties, lib/kernels/sequence_distance_kernel.cu:57-160, and its own tests run
// (5,3,2), (3,5,2) and (3,1,4): tests/test_api.c:59-219): I and D of the last D scores are rings of registers like M -- I[s-E] and
// D[s-E] are what a gap extension reads --, 3 D registers of history instead of D + 2; E == 1 keeps its two registers.
template <int L, int X, int OE, int E, bool BT, bool ASCII>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((BT && E == 1) ? (ASCII ? 5 : 6) : 1, 8)))
wfa_short_kernel(const WfaAlignParams p) {
  static_assert(X >= 1 && X <= 8 && OE >= 1 && OE <= 8 && E >= 1 && E <= OE, "history of eight scores");
  constexpr int D = X > OE ? X : OE;
  constexpr int DG = E == 1 ? 1 : D;      // depth of the I / D history
  extern __shared__ __attribute__((aligned(16))) uint32_t slds[];
  constexpr int G = 64 / L;
  const int lane = threadIdx.x & 63, grp = lane / L, j = lane % L;
  const int cap = p.seq_words_cap;
  // Two sets of staged sequences: the pairs of this iteration and the pairs of the next one (below).
  const size_t set_words = (((size_t)G * 2 * cap) + 3) & ~(size_t)3;
  uint32_t* Pw = slds + (size_t)grp * 2 * cap;
  uint32_t* Pn = Pw + set_words;
  // BT: the group's rows of origin bytes collect in LDS ([score][lane of the group], behind the staged sequences) and go out in
  // 16-byte pieces when the alignment is done.  (A global byte store per score -- four 16-byte pieces of four different cache
  // lines per wave instruction -- cost 0.035 of the 0.26 ms per 100k configs[1] pairs.)
  const int rows_cap = p.max_score + 1;
  uint8_t* const lrows = BT ? reinterpret_cast<uint8_t*>(slds + 2 * set_words) + (size_t)grp * rows_cap * L + j : nullptr;
  uint32_t n_work = p.n_work;
  if (p.n_work_dev) n_work = min(n_work, (uint32_t)*p.n_work_dev);
  const unsigned long long grp_mask = (L == 64) ? ~0ull : (((1ull << L) - 1ull) << (grp * L));
  uint32_t lane_rows = 0;                                   // (group leaders: rows -- score + 1 -- of the alignments they finished; x L = cells)
  // pairs of this wavefront's share that did not leave DONE / went to fail_list / were flagged ALPHABET: wave-uniform counts (scalar
  // registers: the kernel's vector registers decide how many wavefronts a SIMD holds)
  uint32_t not_done = 0, n_fail = 0, n_flag = 0;
  auto count_of = [](const bool pred) { return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pred)); };
  unsigned long long arena_top0 = 0;                        // (the bump pointer as the launch found it: nobody moves it while the launch runs)
  if constexpr (BT) {
    if (p.arena_top_known) {
      // (nobody reads the bump pointer while the launch runs: workgroup 0 moves it past the launch's slots, never beyond the arena)
      arena_top0 = p.arena_top0_value;
      if (blockIdx.x == 0 && lane == 0) {
        const uint32_t tab_units = (uint32_t)(((long long)p.max_score + 2) >> 1), slot = tab_units + (uint32_t)(p.max_score + 1) * (L / 16);
        const unsigned long long t = arena_top0 + (unsigned long long)n_work * slot;
        *p.arena_top = t < p.arena_units ? t : p.arena_units;
      }
    } else arena_top0 = *p.arena_top;
  }

  // Software pipeline over the iterations of a wavefront.  All wavefronts of a launch start together, do the same amount of work
  // and so stay in step: with the loads of an iteration issued when it starts -- work item -> record and status -> sequences,
  // three dependent round trips, every wavefront of the chip asking at once -- the vector pipe sat idle for half of the launch
  // (0.133 ms per 100k configs[1] pairs at 47 % VALU busy; with CIGARs, whose stores the next loads also queued behind, 0.22).
  // Now the work item of iteration i+3, the record of i+2 and the sequences of i+1 are requested at the top of iteration i and
  // looked at one iteration later; the sequences wait in registers (NPF words per lane and sequence; longer ones fetch the
  // rest when they are staged) and go into the other LDS set when iteration i is done.
  constexpr int NPF = ASCII ? 1 : 2;
  struct Meta { uint32_t pair, st, poff, toff; int plen, tlen, budget; uint32_t pasc, tasc; };      // (pasc / tasc: ASCII, dword index of the sequence in the batch)
  struct SeqRegs { uint32_t a[ASCII ? 4 : NPF], b[ASCII ? 4 : NPF]; };
  const uint32_t* const asc = reinterpret_cast<const uint32_t*>(p.ascii);
  uint32_t* const packed_out = const_cast<uint32_t*>(p.packed);
  const uint32_t stride = gridDim.x * G;
  auto fetch_pair = [&](const uint32_t b) -> uint32_t {     // (groups beyond the list look at its last item and stay inactive)
    const uint32_t w = min(b + (uint32_t)grp, n_work - 1u);
    return p.work ? p.work[w] : w;
  };
  auto fetch_meta = [&](const uint32_t pair) -> Meta {
    Meta m;
    m.pair = pair;
    m.st = p.only_pending ? p.status[pair] : (uint32_t)WFA_ST_PENDING;
    const WfaSeqPair mp = p.meta[pair];
    m.plen = (int)mp.pattern_len; m.tlen = (int)mp.text_len;
    // (the packed words: read, or -- ASCII, CIGAR launches -- written)
    m.poff = (ASCII && !BT) ? 0u : (uint32_t)(mp.pattern_offset_packed >> 2); m.toff = (ASCII && !BT) ? 0u : (uint32_t)(mp.text_offset_packed >> 2);
    m.pasc = ASCII ? (uint32_t)(mp.pattern_offset >> 2) : 0u; m.tasc = ASCII ? (uint32_t)(mp.text_offset >> 2) : 0u;
    if (p.budget) m.budget = p.budget[pair];
    else if (p.budget_q > 0) {
      // (k_budget's rule, wfa_host.hip)
      const unsigned long long ql = (unsigned long long)p.budget_q * (unsigned)max(m.plen, m.tlen);
      const long long v = (long long)((p.budget_margin == 100 ? ql : ql * (unsigned)p.budget_margin / 100ull) >> 10) + p.budget_slack;
      m.budget = (int)min(0x3FFFFFFFll, v);
    } else m.budget = p.max_score;
    return m;
  };
  auto words_of = [](const int len) { return ((len + 15) >> 4) + 1; };
  auto seq_ok = [&](const Meta& m, const uint32_t b) {      // will this group stage sequences for the iteration at list position b?
    return b + (uint32_t)grp < n_work && m.st == (uint32_t)WFA_ST_PENDING && words_of(m.plen) <= cap && words_of(m.tlen) <= cap;
  };
  auto fetch_seq = [&](const Meta& m, const bool ok) -> SeqRegs {
    SeqRegs r;
    const int pw = ok ? words_of(m.plen) : 0, tw = ok ? words_of(m.tlen) : 0;
    if constexpr (ASCII) {
      // the sixteen bytes of word j of either sequence (indices clamped to the sequence's last dword: pack_device.h; an empty
      // sequence may sit at the very end of the buffer: its fully masked loads go to the record array)
      const uint32_t* const ps = m.plen ? asc + m.pasc : reinterpret_cast<const uint32_t*>(p.meta);
      const uint32_t* const ts = m.tlen ? asc + m.tasc : reinterpret_cast<const uint32_t*>(p.meta);
      wfa_pack::PackWord a{0u, 0u, 0u, 0u}, b{0u, 0u, 0u, 0u};
      if (j < pw) a = wfa_pack::load_word(ps, (uint32_t)m.plen, (uint32_t)j);
      if (j < tw) b = wfa_pack::load_word(ts, (uint32_t)m.tlen, (uint32_t)j);
      r.a[0] = a.a0; r.a[1] = a.a1; r.a[2] = a.a2; r.a[3] = a.a3;
      r.b[0] = b.a0; r.b[1] = b.a1; r.b[2] = b.a2; r.b[3] = b.a3;
    } else {
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int i = j + u * L;
      r.a[u] = i < pw ? p.packed[(size_t)m.poff + i] : 0u;
      r.b[u] = i < tw ? p.packed[(size_t)m.toff + i] : 0u;
    }
    }
    return r;
  };
  // (ASCII: marks a group whose pair has a byte outside ACGT -- status ALPHABET, counted, out of this kernel's hands)
  auto stage_seq = [&](uint32_t* const Pb, Meta& m, const SeqRegs& r, const bool ok) {
    uint32_t* const Tb = Pb + cap;
    const int pw = ok ? words_of(m.plen) : 0, tw = ok ? words_of(m.tlen) : 0;
    if constexpr (ASCII) {
      uint32_t bad = 0;
      const uint32_t* const ps = m.plen ? asc + m.pasc : reinterpret_cast<const uint32_t*>(p.meta);
      const uint32_t* const ts = m.tlen ? asc + m.tasc : reinterpret_cast<const uint32_t*>(p.meta);
      // (the words go to the packed buffer as well where a backtrace kernel will look for them: CIGAR launches.  A pair that leaves a
      // score-only launch unfinished is packed again by the kernel that takes it over.)
      if (j < pw) { const uint32_t w = wfa_pack::pack_word(wfa_pack::PackWord{r.a[0], r.a[1], r.a[2], r.a[3]}, (uint32_t)m.plen, (uint32_t)j, bad); Pb[j] = w; if constexpr (BT) packed_out[(size_t)m.poff + j] = w; }
      if (j < tw) { const uint32_t w = wfa_pack::pack_word(wfa_pack::PackWord{r.b[0], r.b[1], r.b[2], r.b[3]}, (uint32_t)m.tlen, (uint32_t)j, bad); Tb[j] = w; if constexpr (BT) packed_out[(size_t)m.toff + j] = w; }
#pragma nounroll
      for (int i = j + L; i < pw; i += L) { const uint32_t w = wfa_pack::pack_word(wfa_pack::load_word(ps, (uint32_t)m.plen, (uint32_t)i), (uint32_t)m.plen, (uint32_t)i, bad); Pb[i] = w; if constexpr (BT) packed_out[(size_t)m.poff + i] = w; }
#pragma nounroll
      for (int i = j + L; i < tw; i += L) { const uint32_t w = wfa_pack::pack_word(wfa_pack::load_word(ts, (uint32_t)m.tlen, (uint32_t)i), (uint32_t)m.tlen, (uint32_t)i, bad); Tb[i] = w; if constexpr (BT) packed_out[(size_t)m.toff + i] = w; }
      const bool flagged = (__builtin_amdgcn_ballot_w64(bad != 0u) & grp_mask) != 0ull;
      if (flagged) {
        // WFA2 compares raw bytes: this pair belongs to the byte-compare class, which runs after the packed one
        if (j == 0) {
          atomicAdd(p.n_raw, 1ull);
          p.score[m.pair] = -1;
          p.status[m.pair] = WFA_ST_ALPHABET;
          if (p.cells) p.cells[m.pair] = 0;
        }
        m.st = WFA_ST_ALPHABET;
      }
      n_flag += count_of(flagged && j == 0);
      return;
    }
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int i = j + u * L;
      if (i < pw) Pb[i] = r.a[u];
      if (i < tw) Tb[i] = r.b[u];
    }
#pragma nounroll
    for (int i = j + NPF * L; i < pw; i += L) Pb[i] = p.packed[(size_t)m.poff + i];
#pragma nounroll
    for (int i = j + NPF * L; i < tw; i += L) Tb[i] = p.packed[(size_t)m.toff + i];
  };

  uint32_t base = blockIdx.x * G;
  Meta m0 = {}, m1 = {};
  uint32_t pair2 = 0;
  if (base < n_work) {
    // fill the pipeline
    m0 = fetch_meta(fetch_pair(base));
    m1 = fetch_meta(fetch_pair(base + stride));
    pair2 = fetch_pair(base + 2u * stride);
    const bool ok0 = seq_ok(m0, base);
    const SeqRegs r0 = fetch_seq(m0, ok0);
    stage_seq(Pw, m0, r0, ok0);
  }
  for (; base < n_work; base += stride) {
    // ---- requests for the iterations to come
    const Meta m2 = fetch_meta(pair2);
    const bool ok1 = seq_ok(m1, base + stride);
    const SeqRegs r1 = fetch_seq(m1, ok1);
    const uint32_t pair3 = fetch_pair(base + 3u * stride);
    // ---- this iteration
    const uint32_t w = base + grp;
    const bool active = w < n_work && m0.st == (uint32_t)WFA_ST_PENDING;
    const uint32_t pair = m0.pair;
    const int plen = active ? m0.plen : 0, tlen = active ? m0.tlen : 0;
    uint32_t* const Tw = Pw + cap;
    const int kend = tlen - plen;
    const int pwords = ((plen + 15) >> 4) + 1, twords = ((tlen + 15) >> 4) + 1;
    // the pair's score budget and the diagonal window that can hold an alignment within it (align_kernel.hip: a path that
    // visits diagonal k beyond both 0 and kend pays one gap out and one gap back)
    int budget = p.max_score;
    if (active) budget = min(budget, m0.budget);
    {
      const long long worst = (long long)X * min(plen, tlen) + (kend ? OE + (long long)E * (abs(kend) - 1) : 0);
      budget = (int)min((long long)budget, worst);
    }
    uint32_t status = WFA_ST_DONE;
    int wlo = 0;
    if (active) {
      const long long S = budget, o = OE - E;
      const int ak = kend < 0 ? -kend : kend;
      const bool feasible = (ak ? o + (long long)ak * E : 0) <= S;
      const long long a_hi = S - 2 * o + (long long)kend * E, a_lo = S - 2 * o - (long long)kend * E;
      const int kmax0 = max(0, kend), kmin0 = min(0, kend);
      int whi = a_hi >= 0 ? max(kmax0, (int)min((long long)tlen, a_hi / (2 * E))) : kmax0;
      wlo = a_lo >= 0 ? min(kmin0, -(int)min((long long)plen, a_lo / (2 * E))) : kmin0;
      whi = min(whi, tlen); wlo = max(wlo, -plen);
      if (!feasible) status = WFA_ST_SCORE;
      else if (whi - wlo + 1 > L || pwords > cap || twords > cap) status = WFA_ST_BAND;
    }
    // CIGARs: the row table (8 bytes per score up to the budget, in 16-byte units) and one row of L origin bytes per score
    // Arena space WITHOUT atomics: work item w owns the units [top + w * slot, top + (w + 1) * slot) above the arena's bump pointer
    // as the launch found it, slot = what the launch's largest budget needs (wfa_short_bt_slot_units); the host moves the bump
    // pointer past the whole region with a one-thread kernel behind this launch.  (One returning atomic per alignment on the
    // one bump word -- ~88 per microsecond on this chip -- was 0.3 of the 0.43 ms of a 100k-pair launch.)
    uint32_t tab_unit = WFA_ROW_NONE, rows_unit = 0;
    if constexpr (BT) {
      const uint32_t tab_units = (uint32_t)(((long long)p.max_score + 2) >> 1), slot = tab_units + (uint32_t)(p.max_score + 1) * (L / 16);
      const unsigned long long b = arena_top0 + (unsigned long long)w * slot;
      if (active && status == WFA_ST_DONE) {
        if (b + slot <= p.arena_units) { tab_unit = (uint32_t)b; rows_unit = tab_unit + tab_units; }
        else status = WFA_ST_NOMEM;
      }
    }
    const bool run = active && status == WFA_ST_DONE;
    const int k = wlo + j;                                  // this lane's diagonal, for the whole alignment
    // An offset on diagonal k is inside the matrix -- !(h > tlen || v > plen), h >= 0, v = h - k >= 0: wavefront_compute_affine.c:45-87
    // -- iff max(0, k) <= h <= min(tlen, plen + k): ONE unsigned compare against per-lane constants.  (Two compares and an
    // s_and_b64 left the select behind them with a mask no vector instruction in flight had written: a v_cndmask_b32 that
    // issues once per 23 cycles on this chip instead of 4 -- profiles/r04/valu_classes.txt.)
    const int h_hi = min(tlen, plen + k);
    const int h_lo = h_hi >= max(0, k) ? max(0, k) : INT_MAX / 2;      // (a diagonal that misses the matrix: nothing is inside)
    const uint32_t h_span = (uint32_t)max(h_hi - h_lo, 0);
    __builtin_amdgcn_wave_barrier();                        // (one wavefront: LDS operations execute in order)

    // run length from (v, h) on this lane's diagonal, 16 bases per step, wave-uniform continuation
    auto extend = [&](int h, bool ok) -> int {
      const int hmax = min(plen + k, tlen);
      const int v = h - k;
      int rem = ok ? hmax - h : 0;
      const uint32_t* pw = Pw + (ok ? (v >> 4) : 0);
      const uint32_t* tw = Tw + (ok ? (h >> 4) : 0);
      const uint32_t sa = (uint32_t)v << 1, sb = (uint32_t)h << 1;
      // (32 bases per LDS round trip: the loop goes on while ANY lane of the wavefront still matches -- the longest run of four
      // alignments' main diagonals, five or six 16-base steps per score on 150 bp reads at 2 % -- and every step is a dependent
      // LDS read; the waves of this kernel wait, they do not compete for the vector pipe)
      while (__builtin_amdgcn_ballot_w64(rem > 0) != 0ull) {
        const uint32_t p0 = pw[0], p1 = pw[1], p2 = pw[2], t0 = tw[0], t1 = tw[1], t2 = tw[2];
        const uint32_t d0 = __builtin_amdgcn_alignbit(p1, p0, sa) ^ __builtin_amdgcn_alignbit(t1, t0, sb);
        const uint32_t d1 = __builtin_amdgcn_alignbit(p2, p1, sa) ^ __builtin_amdgcn_alignbit(t2, t1, sb);
        const int run32 = d0 ? (__builtin_ctz(d0) >> 1) : 16 + (d1 ? (__builtin_ctz(d1) >> 1) : 16);
        const int n = min(run32, max(rem, 0));
        h += n;
        rem = (n == 32) ? rem - 32 : 0;
        pw += 2; tw += 2;
      }
      return h;
    };

    // score 0
    int m[D];                                               // M of the scores with (s % D) == index
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = S_NULL;
    int ig[DG], dg[DG];                                     // E == 1: I and D of the last score; else of the scores with (s % D) == index
#pragma unroll
    for (int i = 0; i < DG; ++i) { ig[i] = S_NULL; dg[i] = S_NULL; }
    {
      const bool mine = run && k == 0;
      const int h = extend(0, mine);
      m[0] = mine ? h : S_NULL;
      if constexpr (BT) { if (run) lrows[0] = 0; }
    }
    bool fin = !run;                                        // this lane's group has its result (or never ran)
    int score = -1;
    {
      const bool hit = run && k == kend && m[0] >= tlen;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
      if (run && (bal & grp_mask) != 0ull) { fin = true; score = 0; }
    }
    int s = 0;
    // one score: reads M[s - X], M[s - OE], I[s - E], D[s - E] from their registers (by value: one of them may be the register written)
    auto step = [&](int& m_out, int& i1, int& d1, const int m_x, const int m_o, const int i_src, const int d_src) {
      ++s;
      const int m_ol = from_below<L>(m_o, j), i_e = from_below<L>(i_src, j), m_or = from_above<L>(m_o, j), d_e = from_above<L>(d_src, j);
      const int ins = max(m_ol, i_e) + 1;
      const int del = max(m_or, d_e);
      const int mis = m_x + 1;
      // !(h > tlen || v > plen), unsigned so that NULLs fail too (wavefront_compute_affine.c:45-87)
      const bool i_ok = (uint32_t)(ins - h_lo) <= h_span;
      const bool d_ok = (uint32_t)(del - h_lo) <= h_span;
      i1 = i_ok ? ins : S_NULL;
      d1 = d_ok ? del : S_NULL;
      const int mv = max(max(d1, i1), mis);
      if constexpr (BT) {
        // tie-breaks: gap extension over gap open (wavefront_compute_affine.c:135-143,153-161); M: mismatch, then deletion, then
        // insertion (wavefront_backtrace.c:48-59).  A group that has its result stores nothing more (its rows end at its budget).
        if (!fin) {
          const uint32_t code = (i_e >= m_ol ? BT_I_EXT : 0u) | (d_e >= m_or ? BT_D_EXT : 0u) |
                                ((mis == mv) ? BT_M_X : ((d1 == mv) ? BT_M_D : BT_M_I));
          lrows[s * L] = (uint8_t)code;
        }
      }
      const bool ok = !fin & ((uint32_t)(mv - h_lo) <= h_span);
      const int h = extend(mv, ok);
      m_out = ok ? h : S_NULL;
      const bool hit = ok && k == kend && h >= tlen;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
      if (!fin) {
        if ((bal & grp_mask) != 0ull) { fin = true; score = s; }
        else if (s >= budget) { fin = true; status = WFA_ST_SCORE; }     // (the next score would be past the budget)
      }
    };
    // s % D runs 1, 2, .., D - 1, 0, 1, ..: M[s - X] sits in register (s - X) % D, M[s - OE] in (s - OE) % D
    bool more = __builtin_amdgcn_ballot_w64(!fin) != 0ull;
    while (more) {
#pragma unroll
      for (int r = 1; r <= D; ++r) {
        if (more) {
          if constexpr (E == 1) step(m[r % D], ig[0], dg[0], m[(r - X + D) % D], m[(r - OE + D) % D], ig[0], dg[0]);
          else step(m[r % D], ig[r % D], dg[r % D], m[(r - X + D) % D], m[(r - OE + D) % D], ig[(r - E + D) % D], dg[(r - E + D) % D]);
          more = __builtin_amdgcn_ballot_w64(!fin) != 0ull;
        }
      }
    }
    // ---- the sequences of the next iteration: registers -> the other LDS set.  (Before this iteration's results go out: the wait
    // for the loads would otherwise wait for those stores as well -- one counter, in order.)
    stage_seq(Pn, m1, r1, ok1);
    if constexpr (BT) {
      if (run && status == WFA_ST_DONE) {
        // the rows, LDS -> arena: (score + 1) * L bytes in 16-byte pieces (LDS operations of one wavefront execute in order)
        asm volatile("" ::: "memory");
        {
          const uint4* src = reinterpret_cast<const uint4*>(lrows - j);
          uint4* dst = reinterpret_cast<uint4*>(p.arena + (size_t)rows_unit * 16);
          const int n16 = (score + 1) * (L / 16);
          for (int q = j; q < n16; q += L) dst[q] = src[q];
        }
        // the row table: [score] = {unit of the row, its lower diagonal} (what every tier leaves for the backtrace)
        uint2* tab = reinterpret_cast<uint2*>(p.arena + (size_t)tab_unit * 16);
        for (int t = j; t <= score; t += L) tab[t] = make_uint2(rows_unit + (uint32_t)t * (L / 16), (uint32_t)wlo);
        if (j == 0) p.bt_final_row[pair] = tab_unit;
      }
    }
    if (active && j == 0) {
      p.score[pair] = (status == WFA_ST_DONE) ? score : -1;
      p.status[pair] = status;
      const uint32_t cells = (status == WFA_ST_DONE) ? (uint32_t)(max(score, 0) + 1) * (uint32_t)L : 0u;
      if (p.cells) p.cells[pair] = cells;
      lane_rows += (status == WFA_ST_DONE) ? (uint32_t)(max(score, 0) + 1) : 0u;
    }
    {
      const bool failed = active && j == 0 && p.fail_list && (status == WFA_ST_BAND || status == WFA_ST_SCORE);
      if (failed) p.fail_list[atomicAdd(p.fail_count, 1ull)] = pair;      // (rare)
      n_fail += count_of(failed);
      not_done += count_of(j == 0 && w < n_work && !(active && status == WFA_ST_DONE));      // (skipped pairs -- another class's -- included)
    }
    // ---- everything moves up one stage
    m0 = m1; m1 = m2; pair2 = pair3;
    { uint32_t* const t = Pw; Pw = Pn; Pn = t; }
    __builtin_amdgcn_wave_barrier();
  }
  // cells of this wavefront (the group leaders counted theirs)
  unsigned long long blk_cells = lane_rows;
  for (int d = 32; d > 0; d >>= 1) blk_cells += __shfl_down(blk_cells, d);
  blk_cells *= (unsigned)L;
  if (lane == 0) {
    if (p.wave_parts) {
      ulonglong4* const out = reinterpret_cast<ulonglong4*>(p.wave_parts) + blockIdx.x;
      *out = make_ulonglong4(blk_cells, not_done, n_fail, n_flag);
    }
    else if (blk_cells && p.launch_cells) atomicAdd(p.launch_cells, blk_cells);
  }
}


template <int L, int X, int OE, int E, bool BT, bool ASCII>
void launch_short(const WfaAlignParams& p, int grid, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  const size_t lds = wfa_short_lds_bytes(p, L, BT);
  wfa_launch_timed(wfa_short_kernel<L, X, OE, E, BT, ASCII>, dim3(grid), dim3(64), lds, stream, ev0, ev1, p);
}

// wavefronts of this instantiation a CU holds (registers and LDS)
template <int L, int X, int OE, int E, bool BT, bool ASCII>
int occ_short(size_t lds) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(wfa_short_kernel<L, X, OE, E, BT, ASCII>), 64, lds) != hipSuccess) { nb = 0; (void)hipGetLastError(); }
  return nb;
}

// [x - 1][oe - 1], one table per group width, mode and input (packed words / ASCII); sets with o + e < e do not exist (null entries)
template <int L, int E, bool BT, bool ASCII, int I> constexpr WfaShortEntry short_entry() {
  if constexpr (I % 8 + 1 >= E) return {&launch_short<L, I / 8 + 1, I % 8 + 1, E, BT, ASCII>, &occ_short<L, I / 8 + 1, I % 8 + 1, E, BT, ASCII>};
  else return {nullptr, nullptr};
}
template <int L, int E, bool BT, bool ASCII, int... Is> constexpr std::array<WfaShortEntry, 64> short_table(std::integer_sequence<int, Is...>) { return {short_entry<L, E, BT, ASCII, Is>()...}; }
// [ASCII][BT][L == 32]
template <int E> struct ShortTables {
  static const WfaShortEntry* pick(int ascii, int bt, int l32, int idx) {
    static const std::array<WfaShortEntry, 64> t[2][2][2] = {
        {{short_table<16, E, false, false>(std::make_integer_sequence<int, 64>{}), short_table<32, E, false, false>(std::make_integer_sequence<int, 64>{})},
         {short_table<16, E, true, false>(std::make_integer_sequence<int, 64>{}), short_table<32, E, true, false>(std::make_integer_sequence<int, 64>{})}},
        {{short_table<16, E, false, true>(std::make_integer_sequence<int, 64>{}), short_table<32, E, false, true>(std::make_integer_sequence<int, 64>{})},
         {short_table<16, E, true, true>(std::make_integer_sequence<int, 64>{}), short_table<32, E, true, true>(std::make_integer_sequence<int, 64>{})}}};
    return &t[ascii][bt][l32][idx];
  }
};

}  // namespace
