// Backtrace + CIGAR emission on the GPU.
//
// Replaces, on the device, what the reference does on the host:
//   * the prev-pointer walk over piggy-backed backtrace blocks
//     (lib/kernels/sequence_alignment_kernel.cu:659-683) and
//   * recover_cigar_affine / insert_ops (utils/cigar.c:31-61,96-272), which
//     re-extends matches on the ASCII text between recorded operations.
// The output string is what WFA2's cigar_sprint(print_matches=true) prints
// (external/WFA/alignment/cigar.c:394-426): run-length "nM nX nI nD" items.
//
// Two mappings, chosen by the host driver per pass:
//   * short alignments (64 pairs of packed sequences fit a wavefront's share of LDS): ONE LANE per alignment.
//     wfa_walk_kernel follows the origin bytes written by the align kernel backwards (through the alignment's block of tiles,
//     or its row table: bt_block below) and pushes one byte per edit operation; wfa_emit_kernel replays the
//     operations forwards, re-deriving every match run as a longest-common-prefix on the packed sequences -- identical
//     to the extension the forward pass did, so no offsets had to be stored -- into an upper-bound slot of a scratch
//     (big passes; wfa_text_compact_kernel then packs the texts densely) or, for small passes, twice: once to size the
//     text, once to write it;
//   * long alignments: ONE WAVEFRONT per alignment (wfa_trace_wave_kernel): sequences of the pair, op list and text in
//     LDS, origin bytes fetched as tiles by all 64 lanes.
// Scratch and text space come from wave-aggregated bump allocations (one atomic per wavefront), so the text arena is
// dense and can be copied to the host in one piece.
#include <cstdlib>
#include <type_traits>

#include "wfa_device.h"

namespace {

// How the replay reads a packed sequence: words i and i+1 at a time, at positions that only move forward.
//   SeqDirect  the whole sequence is addressable (LDS copy, or global memory)
//   SeqWindow  8 words per lane in LDS, refilled from global memory when the position leaves them: 64 bytes of LDS per
//              alignment instead of both whole sequences, which is what lets the lane-per-alignment emit kernel run at
//              full occupancy (staging whole sequences capped it at 4 wavefronts per CU: 3.9 ms per 1M cfg3 pairs,
//              all of it exposed latency)
struct SeqDirect {
  const uint32_t* w;
  __device__ __forceinline__ uint2 pair(int i) const { return make_uint2(w[i], w[i + 1]); }
};
struct SeqWindow {
  static constexpr int WORDS = 8;
  const uint32_t* g;     // the sequence in global memory
  int nwords;            // words that may be read there
  uint32_t* slot;        // this lane's WORDS words of LDS
  int base;              // word index of slot[0]
  __device__ __forceinline__ uint2 pair(int i) {
    if ((unsigned)(i - base) >= (unsigned)(WORDS - 1)) {
      base = i;
#pragma unroll
      for (int t = 0; t < WORDS; ++t) slot[t] = (i + t < nwords) ? g[i + t] : 0u;
    }
    return make_uint2(slot[i - base], slot[i - base + 1]);
  }
};

template <bool RAW, typename PV, typename TV>
__device__ __forceinline__ int lcp_seq(PV& Pw, TV& Tw, int plen, int tlen, int v, int h) {
  constexpr int SH = RAW ? 2 : 4;
  constexpr int PER = 1 << SH;
  constexpr int BITS = RAW ? 3 : 1;
  int n_total = 0;
  int rem = min(plen - v, tlen - h);
  if constexpr (std::is_same<PV, SeqDirect>::value && std::is_same<TV, SeqDirect>::value) {
    // whole sequences addressable (LDS): THREE words per step -- every step is a dependent LDS round trip, and the runs between
    // the operations of reads at a few per cent of error are two or three words long.  (Trace per batch, BASELINE configs[2] /
    // configs[3], one, two, three, four words per step: 3.30 / 3.20 / 3.15 / 3.15 ms and 2.33 / 2.20 / 2.16 / 2.27 ms.)
    while (rem > 0) {
      const uint32_t* pw = Pw.w + (v >> SH); const uint32_t* tw = Tw.w + (h >> SH);
      const uint32_t p0 = pw[0], p1 = pw[1], p2 = pw[2], p3 = pw[3], t0 = tw[0], t1 = tw[1], t2 = tw[2], t3 = tw[3];
      const uint32_t sa = (uint32_t)(v & (PER - 1)) << BITS, sb = (uint32_t)(h & (PER - 1)) << BITS;
      const uint32_t d0 = __builtin_amdgcn_alignbit(p1, p0, sa) ^ __builtin_amdgcn_alignbit(t1, t0, sb);
      const uint32_t d1 = __builtin_amdgcn_alignbit(p2, p1, sa) ^ __builtin_amdgcn_alignbit(t2, t1, sb);
      const uint32_t d2 = __builtin_amdgcn_alignbit(p3, p2, sa) ^ __builtin_amdgcn_alignbit(t3, t2, sb);
      // (nested, first word first: a chain of selects from the last word down measured 2 % slower)
      int n = d0 ? (__builtin_ctz(d0) >> BITS) : PER + (d1 ? (__builtin_ctz(d1) >> BITS) : PER + (d2 ? (__builtin_ctz(d2) >> BITS) : PER));
      n = min(n, rem);
      n_total += n; h += n; v += n; rem -= n;
      if (n < 3 * PER) break;
    }
    return n_total;
  }
  while (rem > 0) {
    const uint2 pw = Pw.pair(v >> SH), tw = Tw.pair(h >> SH);
    const uint32_t a = __builtin_amdgcn_alignbit(pw.y, pw.x, (v & (PER - 1)) << BITS);
    const uint32_t b = __builtin_amdgcn_alignbit(tw.y, tw.x, (h & (PER - 1)) << BITS);
    const uint32_t d = a ^ b;
    int n = d ? (__builtin_ctz(d) >> BITS) : PER;
    n = min(n, rem);
    n_total += n; h += n; v += n; rem -= n;
    if (n < PER) break;
  }
  return n_total;
}

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
  const uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

// All 64 lanes must call.  Returns this lane's byte offset.
__device__ __forceinline__ unsigned long long wave_alloc(unsigned long long* top, uint32_t need, int lane) {
  uint32_t incl = need;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  const uint32_t total = __shfl(incl, 63);
  unsigned long long base = 0;
  if (lane == 63 && total) base = atomicAdd(top, (unsigned long long)total);
  base = shfl64(base, 63);
  return base + incl - need;
}

// The same for a workgroup of W wavefronts: ONE atomic per workgroup (a returning atomic on the one bump counter costs ~11 ns
// whoever issues it: at a kernel's start every resident wavefront queues for it).  All threads must call; two barriers.
template <int W>
__device__ __forceinline__ unsigned long long block_alloc(unsigned long long* top, uint32_t need, uint32_t* wave_total, unsigned long long* block_base) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = need;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
  if (lane == 63) wave_total[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < W; ++w) tot += wave_total[w];
    *block_base = tot ? atomicAdd(top, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wv; ++w) before += wave_total[w];
  return *block_base + before + incl - need;
}

__device__ __forceinline__ int dec_digits(uint32_t n) {
  return n < 10 ? 1 : n < 100 ? 2 : n < 1000 ? 3 : n < 10000 ? 4 : n < 100000 ? 5 : n < 1000000 ? 6 : n < 10000000 ? 7 :
         n < 100000000 ? 8 : n < 1000000000 ? 9 : 10;
}

struct RleSink {
  char* out;      // nullptr: count only
  uint32_t cap;   // bytes available at `out` (items that would not fit are only counted)
  uint32_t len;
  uint32_t run;
  char op;
  int x, o, e;    // penalties: the gap-affine cost of the emitted items is accumulated while writing
  int cost;
  bool words;     // the slot at `out` is this lane's own up to `cap`: an item may be stored as one (unaligned) 32-bit word
  __device__ __forceinline__ void flush() {
    if (run == 0) return;
    const int nd = dec_digits(run);
    if (out && len + (uint32_t)nd + 2u <= cap) {
      if (words && run < 1000u && len + 4u <= cap) {
        // up to three digits and the operation, assembled in a register (branch-free: the lanes of a wavefront hold
        // runs of every length) and written with one store; the bytes beyond the item belong to the next one
        const uint32_t d2 = run / 100u, r2 = run - 100u * d2, d1 = r2 / 10u, d0 = r2 - 10u * d1;
        const uint32_t w4 = (0x30u + d2) | ((0x30u + d1) << 8) | ((0x30u + d0) << 16) | ((uint32_t)(uint8_t)op << 24);
        const uint32_t w = w4 >> (8u * (uint32_t)(3 - nd));
        __builtin_memcpy(out + len, &w, 4);
      } else {
        uint32_t r = run;
        for (int i = nd - 1; i >= 0; --i) { out[len + i] = (char)('0' + r % 10); r /= 10; }
        out[len + nd] = op;
      }
    }
    // (utils/verification.c:91-146 of the reference: a new gap wherever the operation changes)
    if (op == 'X') cost += x * (int)run;
    else if (op != 'M') cost += o + e * (int)run;
    len += (uint32_t)nd + 1;
    run = 0;
  }
  __device__ __forceinline__ void push(char o, uint32_t n) {
    if (n == 0) return;
    if (o != op) { flush(); op = o; }
    run += n;
  }
};

// (OPS_LDS: the op list is in LDS -- said explicitly, a pointer that was global or LDS by a run-time choice loads with flat_load)
template <bool RAW, bool OPS_LDS = false, typename PV, typename TV>
__device__ __forceinline__ uint32_t replay_views(const uint8_t* ops, uint32_t nops, PV& Pw, TV& Tw,
                                                 int plen, int tlen, char* out, int x, int o, int e, int* cost,
                                                 uint32_t out_cap = 0xFFFFFFFFu, bool words = false) {
  RleSink sink{out, out_cap, 0, 0, 0, x, o, e, 0, words};
  int v = 0, h = 0;
  int n = lcp_seq<RAW>(Pw, Tw, plen, tlen, v, h);
  sink.push('M', (uint32_t)n); v += n; h += n;
  // The op list is consumed four ops per aligned 32-bit load: with one byte load per step every iteration of every lane
  // waited a full round trip to L2 (that, not the LCPs, was most of the emit kernel's time).
  // (the aligned pointer by pointer arithmetic: through an integer it loses its address space and every load becomes a flat_load)
  const uint32_t skip = (uint32_t)(reinterpret_cast<uintptr_t>(ops) & 3);
  const uint32_t* ops4 = reinterpret_cast<const uint32_t*>(ops - skip);
  auto op_word = [&](uint32_t idx) -> uint32_t {
    if constexpr (OPS_LDS) return ((const __attribute__((address_space(3))) uint32_t*)ops4)[idx];
    else return ops4[idx];
  };
  uint32_t word = nops ? (op_word(0) >> (8 * skip)) : 0u;
  for (uint32_t i = 0; i < nops; ++i) {
    const uint32_t op = word & 0xFFu;
    if (((i + 1 + skip) & 3u) == 0u) { if (i + 1 < nops) word = op_word((i + 1 + skip) >> 2); } else word >>= 8;
    {
      // (one push for the three kinds of operation: the lanes of a wavefront hold all of them at once)
      const uint32_t k3 = op & 3u;
      sink.push(k3 == OP_X ? 'X' : (k3 == OP_I ? 'I' : 'D'), 1);
      v += (k3 != OP_I) ? 1 : 0; h += (k3 != OP_D) ? 1 : 0;
    }
    if (op & OP_EXT_AFTER) {
      n = lcp_seq<RAW>(Pw, Tw, plen, tlen, v, h);
      sink.push('M', (uint32_t)n); v += n; h += n;
    }
  }
  sink.flush();
  if (out && sink.len < out_cap) out[sink.len] = '\0';
  if (cost) *cost = sink.cost;
  // a consistent trace ends exactly at the corner
  return (v == plen && h == tlen) ? sink.len : 0xFFFFFFFFu;
}

template <bool RAW, bool OPS_LDS = false>
__device__ __forceinline__ uint32_t replay(const uint8_t* ops, uint32_t nops, const uint32_t* Pw, const uint32_t* Tw,
                                           int plen, int tlen, char* out, int x, int o, int e, int* cost,
                                           uint32_t out_cap = 0xFFFFFFFFu, bool words = false) {
  SeqDirect pv{Pw}, tv{Tw};
  return replay_views<RAW, OPS_LDS>(ops, nops, pv, tv, plen, tlen, out, x, o, e, cost, out_cap, words);
}

constexpr int TRACE_THREADS = 64;

// Where the origin bytes of a pair are (wfa_device.h): behind a row table -- [score] = {arena unit of the row, lo} --, or (the one-wave
// exact tier, round 6) in the pair's block of TILES: a 64-byte header {WFA_ROW_NONE, wlo, tile columns, budget}, then 64-byte tiles of
// 4 scores x 16 diagonals, tile (s >> 2, (k - wlo) >> 4), byte (s & 3) * 16 + ((k - wlo) & 15).
struct BtBlock { const uint8_t* blk; bool tiled; int wlo; uint32_t cols; };
__device__ __forceinline__ BtBlock bt_block(const WfaTraceParams& p, const uint32_t pair) {
  BtBlock b;
  b.blk = p.arena + (size_t)p.bt_final_row[pair] * 16;
  const uint4 hdr = *reinterpret_cast<const uint4*>(b.blk);
  b.tiled = hdr.x == WFA_ROW_NONE; b.wlo = (int)hdr.y; b.cols = hdr.z;
  return b;
}
// The wave- and group-per-alignment kernels read origin bytes through a tile in LDS, one row of it per lane: TILE_LDS_W bytes.  Row of
// score sr around diagonal k into dst; kbase: the diagonal of dst[0].  Row-table layout: 16 bytes, diagonals k-8 .. k+7 (one unaligned
// load).  Tiles: the two 16-diagonal tile columns whose 32 diagonals have k at least 8 from either end (two aligned 16-byte loads).
// Every lane sets the same kbase (it depends on k and the pair only).  Rows of scores < 0 and bytes outside the block read as 0.
constexpr int TILE_LDS_W = 32;
__device__ __forceinline__ void fetch_bt_rows(const WfaTraceParams& p, const BtBlock& bb, const int sr, const int k, int& kbase, uint8_t* dst) {
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
  if (bb.tiled) {
    const int c0 = ((k - bb.wlo) - 8) >> 4;      // (arithmetic shift: -1 for diagonals 0..7 of the window)
    kbase = bb.wlo + c0 * 16;
    if (sr >= 0) {
      const uint8_t* const row = bb.blk + 64u + (size_t)((uint32_t)sr >> 2) * bb.cols * 64u + ((uint32_t)sr & 3u) * 16u;
      if (c0 >= 0 && (uint32_t)c0 < bb.cols) v0 = *reinterpret_cast<const uint4*>(row + (size_t)c0 * 64u);
      if (c0 + 1 >= 0 && (uint32_t)(c0 + 1) < bb.cols) v1 = *reinterpret_cast<const uint4*>(row + (size_t)(c0 + 1) * 64u);
    }
  } else {
    kbase = k - 8;
    if (sr >= 0) {
      const uint2 row = reinterpret_cast<const uint2*>(bb.blk)[sr];
      // (entries of scores without a wavefront are stale: their cells are never consulted, but the address must be a safe one)
      const long long off = (long long)row.x * 16 + ((long long)kbase - (int)row.y);
      if (off >= 0 && (unsigned long long)off + 16 <= p.arena_bytes) {
        struct __attribute__((packed, aligned(1))) U16 { uint32_t w[4]; };
        const U16 t = *reinterpret_cast<const U16*>(p.arena + off);
        v0 = make_uint4(t.w[0], t.w[1], t.w[2], t.w[3]);
      }
    }
  }
  reinterpret_cast<uint4*>(dst)[0] = v0;
  reinterpret_cast<uint4*>(dst)[1] = v1;
}

// The backward walk of ONE alignment (one lane): follows the origin bytes from the final cell to (0, 0) and leaves the operations
// as a byte list that grows downwards from q_end (op number n, 0 = last operation of the alignment, is byte q_end[-1-n]; q_end is
// 4-byte aligned, four ops go out as one word).  tab_cache: this lane's 8 row-table entries of LDS.  false: broken trace.
__device__ __forceinline__ bool walk_ops(const WfaTraceParams& p, const uint32_t pair, const int score, const int plen, const int tlen,
                                         uint8_t* const q_end, const uint32_t need_ops, uint2* const tab_cache_lane, uint32_t& nops_out) {
  bool fail = false;
  uint32_t nops = 0, word = 0;
  {
    // row table of the pair: [score] = {arena unit of the origin bytes, lo} -- or (the one-wave exact tier, round 6) the pair's block of
    // TILES: a 64-byte header {WFA_ROW_NONE, wlo, tile columns, budget}, then 64-byte tiles of 4 scores x 16 diagonals, tile
    // (s >> 2, (k - wlo) >> 4), byte (s & 3) * 16 + ((k - wlo) & 15): no table to consult, and the ~55 origin bytes of a 1 kbp
    // alignment sit on ~35 cache lines instead of 55 + 19 lines of row table
    const uint8_t* const blk = p.arena + (size_t)p.bt_final_row[pair] * 16;
    const uint2* tab = reinterpret_cast<const uint2*>(blk);
    const uint4 hdr = *reinterpret_cast<const uint4*>(blk);
    const bool tiled = hdr.x == WFA_ROW_NONE;
    const int t_wlo = (int)hdr.y; const uint32_t t_cols = hdr.z;
    int k = tlen - plen, s = score;
    int state = 0;  // 0: M, 1: I, 2: D
    int cached = -1;
    while (s > 0) {
      if (nops >= need_ops) { fail = true; break; }
      uint32_t code;
      if (tiled) {
        // (the tile is kept in this lane's 64 bytes of LDS -- the row-table cache of the other layout -- until the walk leaves it: a
        // mismatch at x = 2 stays in its tile every other step)
        const uint32_t d = (uint32_t)(k - t_wlo);
        if ((d >> 4) >= t_cols) { fail = true; break; }      // (a broken trace must not leave the block)
        const int tile_id = (int)(((uint32_t)s >> 2) * t_cols + (d >> 4));
        if (tile_id != cached) {
          cached = tile_id;
          const uint4* src = reinterpret_cast<const uint4*>(blk + 64u + (size_t)tile_id * 64u);
          const uint4 a = src[0], b = src[1], c2 = src[2], d4 = src[3];
          tab_cache_lane[0] = make_uint2(a.x, a.y); tab_cache_lane[1] = make_uint2(a.z, a.w);
          tab_cache_lane[2] = make_uint2(b.x, b.y); tab_cache_lane[3] = make_uint2(b.z, b.w);
          tab_cache_lane[4] = make_uint2(c2.x, c2.y); tab_cache_lane[5] = make_uint2(c2.z, c2.w);
          tab_cache_lane[6] = make_uint2(d4.x, d4.y); tab_cache_lane[7] = make_uint2(d4.z, d4.w);
        }
        code = reinterpret_cast<const uint8_t*>(tab_cache_lane)[((uint32_t)s & 3u) * 16u + (d & 15u)];
      } else {
        if ((s >> 3) != cached) {
          cached = s >> 3;
          const uint4* src = reinterpret_cast<const uint4*>(tab + (cached << 3));
          const uint4 a = src[0], b = src[1], c2 = src[2], d = src[3];
          tab_cache_lane[0] = make_uint2(a.x, a.y); tab_cache_lane[1] = make_uint2(a.z, a.w);
          tab_cache_lane[2] = make_uint2(b.x, b.y); tab_cache_lane[3] = make_uint2(b.z, b.w);
          tab_cache_lane[4] = make_uint2(c2.x, c2.y); tab_cache_lane[5] = make_uint2(c2.z, c2.w);
          tab_cache_lane[6] = make_uint2(d.x, d.y); tab_cache_lane[7] = make_uint2(d.z, d.w);
        }
        const uint2 row = tab_cache_lane[s & 7];
        code = p.arena[(size_t)row.x * 16 + (uint32_t)(k - (int)row.y)];
      }
      uint32_t op;
      if (state == 0) {
        const uint32_t org = code & BT_M_MASK;
        if (org == BT_M_X) { op = OP_X | OP_EXT_AFTER; s -= p.x; }
        else if (org == BT_M_I) {
          op = OP_I | OP_EXT_AFTER; --k;
          if (code & BT_I_EXT) { s -= p.e; state = 1; } else { s -= p.oe; }
        } else if (org == BT_M_D) {
          op = OP_D | OP_EXT_AFTER; ++k;
          if (code & BT_D_EXT) { s -= p.e; state = 2; } else { s -= p.oe; }
        } else { fail = true; break; }
      } else if (state == 1) {
        op = OP_I; --k;
        if (code & BT_I_EXT) { s -= p.e; } else { s -= p.oe; state = 0; }
      } else {
        op = OP_D; ++k;
        if (code & BT_D_EXT) { s -= p.e; } else { s -= p.oe; state = 0; }
      }
      // the list grows downwards from q_end: op number n (0 = last operation of the alignment) is byte q_end[-1-n]
      word = (word << 8) | op;
      ++nops;
      if ((nops & 3u) == 0u) reinterpret_cast<uint32_t*>(q_end)[-(int)(nops >> 2)] = word;
    }
    if (s != 0 || state != 0 || k != 0) fail = true;
    if (!fail && (nops & 3u)) {
      // the last, partial word: its ops are the first operations of the alignment
      uint8_t* qq = q_end - (nops & ~3u);
      for (int r = (int)(nops & 3u) - 1; r >= 0; --r) *--qq = (uint8_t)(word >> (8 * r));     // (oldest of them first: highest address)
    }
    }
  nops_out = nops;
  return !fail;
}

// Phase 1 kernel: backward walk over the origin bytes, one lane per alignment: a chain of dependent HBM reads.  Leaves the op
// list of each pair in the scratch arena (ops_off/nops per pair).
constexpr int WALK_WAVES = 4;      // (wavefronts per workgroup: they share the allocation of their op lists, block_alloc)
__global__ void __launch_bounds__(WALK_WAVES * 64) wfa_walk_kernel(const WfaTraceParams p) {
  // The walk is bound by memory TRANSACTIONS (one dependent 64-byte access per step, half a million lanes in flight: no
  // line survives in L2 until its next use), not by bytes.  Per operation it needs a row-table entry, an origin byte and
  // an op store; two of the three are batched: row-table entries are fetched 8 at a time (one 64-byte line) into this
  // lane's LDS slot, ops are collected four to a register and stored as one word.
  __shared__ uint2 tab_cache[WALK_WAVES * 64][9];          // [thread][entry & 7] (9: bank spread)
  __shared__ uint32_t wave_total[WALK_WAVES];
  __shared__ unsigned long long block_base;
  // A grid of two workgroups (eight wavefronts) per CU strides over the list (WfaTraceParams::walk_grid_cap): HBM serves no more
  // random 64-byte accesses with 32 wavefronts per CU than with 8, and with every wave slot taken the time of the kernel fell
  // into one of two modes by process (1M x 1 kbp, trace ms per batch at 32 / 16 / 12 / 8 wavefronts per CU: 3.14-3.32 /
  // 3.17-3.25 / 3.09-3.25 / 3.17 every time -- profiles/r04/trace_slices.txt).
  for (uint32_t first = blockIdx.x * (WALK_WAVES * 64); first < p.n_work; first += gridDim.x * (WALK_WAVES * 64)) {
    const uint32_t gid = first + threadIdx.x;
    bool active = gid < p.n_work;
    uint32_t pair = 0;
    if (active) pair = p.work ? p.work[gid] : gid;
    if (active && p.status[pair] != WFA_ST_DONE) active = false;
    if (active && p.bt_final_row[pair] == WFA_ROW_NONE) active = false;      // (walked by its wavefront kernel: the op list is in the arena)
    int score = 0, plen = 0, tlen = 0;
    if (active) {
      score = p.score[pair];
      plen = (int)p.meta[pair].pattern_len; tlen = (int)p.meta[pair].text_len;
    }
    // every operation costs at least min(x, e) >= 1, so `score` bytes suffice
    const uint32_t need_ops = active ? (((uint32_t)score + 3u) & ~3u) : 0u;
    const unsigned long long ops_off = block_alloc<WALK_WAVES>(p.ops_top, need_ops, wave_total, &block_base);
    bool fail = active && (ops_off + need_ops > p.ops_cap);
    uint8_t* const q_begin = p.ops + ops_off;
    uint8_t* const q_end = q_begin + need_ops;       // (4-byte aligned: the allocations are multiples of 4)
    uint32_t nops = 0;
    if (active && !fail) fail = !walk_ops(p, pair, score, plen, tlen, q_end, need_ops, tab_cache[threadIdx.x], nops);
    if (active) {
      // the op list now sits at [q_end - nops, q_end); cigar_off/cigar_len carry it to the emit kernel
      p.cigar_off[pair] = (unsigned long long)(q_end - nops - p.ops);
      p.cigar_len[pair] = fail ? 0xFFFFFFFFu : nops;
    }
    __syncthreads();      // (block_alloc's shared words are written again by the next round)
  }
}

// Forward replay with 8-word LDS windows (SeqWindow): used when whole sequences do not fit and the wave-per-alignment
// kernel is switched off (WFAGPU_NO_WAVE_TRACE), or for A/B runs (WFAGPU_EMIT_WINDOW); two replays.
__global__ void __launch_bounds__(TRACE_THREADS) wfa_emit_win_kernel(const WfaTraceParams p) {
  __shared__ uint32_t win_lds[64 * (2 * SeqWindow::WORDS + 1)];
  const uint32_t gid = blockIdx.x * TRACE_THREADS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool active = gid < p.n_work;
  uint32_t pair = 0;
  if (active) pair = p.work ? p.work[gid] : gid;
  if (active && p.status[pair] != WFA_ST_DONE) active = false;
  int plen = 0, tlen = 0;
  const uint8_t* q = nullptr; uint32_t nops = 0;
  bool fail = false;
  uint32_t* slot = win_lds + lane * (2 * SeqWindow::WORDS + 1);     // (odd stride: lanes at the same window index hit different banks)
  SeqWindow pv{nullptr, 0, slot, -1000}, tv{nullptr, 0, slot + SeqWindow::WORDS, -1000};
  if (active) {
    const WfaSeqPair mp = p.meta[pair];
    plen = (int)mp.pattern_len; tlen = (int)mp.text_len;
    const int sh = p.raw ? 2 : 4;
    pv.g = p.packed + ((p.raw ? mp.pattern_offset : mp.pattern_offset_packed) >> 2);
    tv.g = p.packed + ((p.raw ? mp.text_offset : mp.text_offset_packed) >> 2);
    pv.nwords = ((plen + (1 << sh) - 1) >> sh) + 1; tv.nwords = ((tlen + (1 << sh) - 1) >> sh) + 1;
    { const unsigned long long off = p.cigar_off[pair]; q = ((off & WFA_OPS_IN_ARENA) ? p.arena : p.ops) + (off & ~WFA_OPS_IN_ARENA); }
    nops = p.cigar_len[pair];
    fail = nops == 0xFFFFFFFFu;
  }
  uint32_t len = 0;
  if (active && !fail) {
    len = p.raw ? replay_views<true>(q, nops, pv, tv, plen, tlen, nullptr, 0, 0, 0, nullptr)
                : replay_views<false>(q, nops, pv, tv, plen, tlen, nullptr, 0, 0, 0, nullptr);
    if (len == 0xFFFFFFFFu) fail = true;
  }
  const uint32_t need_txt = (active && !fail) ? len + 1u : 0u;
  const unsigned long long txt_off = wave_alloc(p.text_top, need_txt, lane);
  if (active && !fail && txt_off + need_txt > p.text_cap) fail = true;
  if (active) {
    if (!fail) {
      int cost = 0;
      pv.base = -1000; tv.base = -1000;
      if (p.raw) replay_views<true>(q, nops, pv, tv, plen, tlen, p.text + txt_off, p.x, p.oe - p.e, p.e, &cost);
      else replay_views<false>(q, nops, pv, tv, plen, tlen, p.text + txt_off, p.x, p.oe - p.e, p.e, &cost);
      p.cigar_off[pair] = txt_off;
      p.cigar_len[pair] = len;
      // (the text's own gap-affine cost must equal the score; adaptive band: the score follows the text -- see below)
      if (cost != p.score[pair]) {
        if (p.score_fix) p.score_fix[pair] = cost;
        else p.cigar_len[pair] = 0xFFFFFFFFu;
      }
    } else {
      p.cigar_off[pair] = 0;
      p.cigar_len[pair] = 0xFFFFFFFFu;
    }
  }
}

// The packed sequences of the wavefront's first PPW pairs -> LDS (lane j's pair at seq_lds + j * seq_lds_stride: pattern words,
// then text words), coalesced: the whole wavefront copies one pair after the other, eight pairs per round with all sixteen loads
// in flight before the first LDS store (one pair per iteration waited a full memory round trip per load: 128 of them in a row,
// half of the emit kernel's time).  Pw / Tw come in as the lanes' global addresses and leave as their LDS addresses.
__device__ __forceinline__ void stage_pairs(const WfaTraceParams& p, const int PPW, const int lane, const bool active, const int plen, const int tlen,
                                            const uint32_t*& Pw, const uint32_t*& Tw, uint32_t* const seq_lds) {
  const int sh = p.raw ? 2 : 4;
  const int pw = active ? ((plen + (1 << sh) - 1) >> sh) + 1 : 0, tw = active ? ((tlen + (1 << sh) - 1) >> sh) + 1 : 0;
  const int stride = p.seq_lds_stride;   // odd number of words per lane
  constexpr int SG = 8;
  for (int j0 = 0; j0 < PPW; j0 += SG) {      // (PPW is a multiple of SG)
    const uint32_t* gp[SG]; const uint32_t* gt[SG]; int pwj[SG], twj[SG];
    int maxw = 0;
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const int j = j0 + u;     // (uniform: the lanes' values come over as scalars)
      gp[u] = reinterpret_cast<const uint32_t*>(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(reinterpret_cast<unsigned long long>(Pw) >> 32), j) << 32) |
                                                (uint32_t)__builtin_amdgcn_readlane((int)reinterpret_cast<unsigned long long>(Pw), j));
      gt[u] = reinterpret_cast<const uint32_t*>(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(reinterpret_cast<unsigned long long>(Tw) >> 32), j) << 32) |
                                                (uint32_t)__builtin_amdgcn_readlane((int)reinterpret_cast<unsigned long long>(Tw), j));
      pwj[u] = __builtin_amdgcn_readlane(pw, j); twj[u] = __builtin_amdgcn_readlane(tw, j);
      maxw = max(maxw, max(pwj[u], twj[u]));
    }
    for (int i0 = 0; i0 < maxw; i0 += 64) {
      const int i = i0 + lane;
      uint32_t a[SG], b[SG];
      // (pointers rebuilt from two scalars have no address space: said here, or every one of these loads is a flat_load)
      using GlobalWords = const __attribute__((address_space(1))) uint32_t*;
#pragma unroll
      for (int u = 0; u < SG; ++u) { a[u] = i < pwj[u] ? ((GlobalWords)gp[u])[i] : 0u; b[u] = i < twj[u] ? ((GlobalWords)gt[u])[i] : 0u; }
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        uint32_t* dst = seq_lds + (size_t)(j0 + u) * stride;
        if (i < pwj[u]) dst[i] = a[u];
        if (i < twj[u]) dst[pwj[u] + i] = b[u];
      }
    }
  }
  __syncthreads();
  Pw = seq_lds + (size_t)lane * stride;
  Tw = Pw + pw;
}

// SHORT alignments, everything in one kernel (round 4): one lane per alignment; the wavefront stages its pairs' sequences in LDS,
// every lane walks its origin bytes backwards into an op list in LDS (ops_lds_bytes per lane: the chain's largest score, one byte
// per operation), replays it once to size the text, buys the text's place in the dense arena (one atomic per wavefront) and
// replays it again straight into it.  Against walk + emit + compaction: no op list and no text scratch in global memory, one
// allocation instead of three (every one of them a returning atomic per wavefront on one counter: ~11 ns each, 17 us per
// 100k-pair kernel just for those), one launch instead of three.
// Workgroups of LANE_WAVES wavefronts, each with its own pairs and its own share of LDS: what they share is the allocation of their
// texts (block_alloc: one returning atomic per workgroup -- the 1563 wavefronts of a 100k-pair launch, all at that point within
// microseconds of each other, queued ~11 ns each for the one counter: configs[1] with CIGARs, backtrace 0.049 -> 0.037 ms.  The same
// for wfa_emit_kernel, whose wavefronts come by in rounds, changed nothing: 1M x 1 kbp pairs 3.05 / 3.07 ms.)
constexpr int LANE_WAVES = 4;
__host__ __device__ inline size_t lane_kernel_wave_bytes(int emit_pairs, int seq_lds_stride, int ops_lds_bytes) {
  return ((size_t)emit_pairs * seq_lds_stride * 4 + (size_t)emit_pairs * ops_lds_bytes + (size_t)64 * 9 * 8 + 15) & ~(size_t)15;
}
template <bool RAW>
__global__ void __launch_bounds__(LANE_WAVES * 64) wfa_trace_lane_kernel(const WfaTraceParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lane_lds[];
  __shared__ uint32_t wave_total[LANE_WAVES];
  __shared__ unsigned long long block_base;
  const int PPW = p.emit_pairs;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* const seq_lds = lane_lds + (size_t)wv * (lane_kernel_wave_bytes(PPW, p.seq_lds_stride, p.ops_lds_bytes) >> 2);
  const uint32_t gid = (blockIdx.x * (uint32_t)LANE_WAVES + (uint32_t)wv) * (uint32_t)PPW + (uint32_t)lane;
  uint8_t* const ops_lds = reinterpret_cast<uint8_t*>(seq_lds + (size_t)PPW * p.seq_lds_stride);
  uint2* const tab_cache = reinterpret_cast<uint2*>(ops_lds + (size_t)PPW * p.ops_lds_bytes);      // [lane][9]
  bool active = lane < PPW && gid < p.n_work;
  uint32_t pair = 0;
  if (active) pair = p.work ? p.work[gid] : gid;
  if (active && p.status[pair] != WFA_ST_DONE) active = false;
  int plen = 0, tlen = 0, score = 0;
  const uint32_t* Pw = nullptr; const uint32_t* Tw = nullptr;
  if (active) {
    const WfaSeqPair mp = p.meta[pair];
    plen = (int)mp.pattern_len; tlen = (int)mp.text_len;
    Pw = p.packed + ((RAW ? mp.pattern_offset : mp.pattern_offset_packed) >> 2);
    Tw = p.packed + ((RAW ? mp.text_offset : mp.text_offset_packed) >> 2);
    score = p.score[pair];
  }
  stage_pairs(p, PPW, lane, active, plen, tlen, Pw, Tw, seq_lds);
  const uint32_t need_ops = active ? (((uint32_t)score + 3u) & ~3u) : 0u;
  bool fail = active && need_ops > (uint32_t)p.ops_lds_bytes;
  uint8_t* const q_end = ops_lds + (size_t)lane * p.ops_lds_bytes + need_ops;      // (the lane's slot: 4-byte aligned, like need_ops)
  uint32_t nops = 0;
  if (active && !fail) fail = !walk_ops(p, pair, score, plen, tlen, q_end, need_ops, tab_cache + (size_t)lane * 9, nops);
  const uint8_t* const q = q_end - nops;
  uint32_t len = 0;
  if (active && !fail) {
    len = replay<RAW>(q, nops, Pw, Tw, plen, tlen, nullptr, 0, 0, 0, nullptr);
    if (len == 0xFFFFFFFFu) fail = true;
  }
  const uint32_t need_txt = (active && !fail) ? len + 1u : 0u;
  const unsigned long long txt_off = block_alloc<LANE_WAVES>(p.text_top, need_txt, wave_total, &block_base);
  if (active && !fail && txt_off + need_txt > p.text_cap) fail = true;
  if (active) {
    if (!fail) {
      int cost = 0;
      replay<RAW>(q, nops, Pw, Tw, plen, tlen, p.text + txt_off, p.x, p.oe - p.e, p.e, &cost);
      p.cigar_off[pair] = txt_off;
      p.cigar_len[pair] = len;
      // (the text's own gap-affine cost must equal the score; adaptive band: the score follows the text -- see wfa_emit_kernel)
      if (cost != score) {
        if (p.score_fix) p.score_fix[pair] = cost;
        else p.cigar_len[pair] = 0xFFFFFFFFu;
      }
    } else {
      p.cigar_off[pair] = 0;
      p.cigar_len[pair] = 0xFFFFFFFFu;
    }
  }
}

// Forward replay, the default for short alignments: the 64 pairs of a block are first copied into LDS (coalesced, one
// pair at a time by the whole wavefront), so the many small reads of the replay never leave the CU.  (Reading the packed
// sequences straight from global memory made every 4-byte read miss L1 and L2 -- the working set of all resident lanes
// is far larger than both: 57 GB fetched per 1M pairs in round 1.)
template <bool SEQ_LDS, bool OPS_LDS>
__global__ void __launch_bounds__(TRACE_THREADS) wfa_emit_kernel(const WfaTraceParams p) {
  static_assert(SEQ_LDS, "sequences that do not fit LDS go through wfa_emit_win_kernel or wfa_trace_wave_kernel");
  extern __shared__ __attribute__((aligned(16))) uint32_t seq_lds[];
  // (the first PPW lanes of the wavefront carry an alignment each: staging 64 pairs of 1 kbp reads takes 33 KB of LDS, four
  // wavefronts per CU -- one per SIMD, every dependent LDS read of the replay fully exposed; fewer pairs per wavefront, more
  // wavefronts and more lanes per CU)
  const int PPW = p.emit_pairs;
  const int lane = threadIdx.x & 63;
  const uint32_t gid = blockIdx.x * (uint32_t)PPW + (uint32_t)lane;
  bool active = lane < PPW && gid < p.n_work;
  uint32_t pair = 0;
  if (active) pair = p.work ? p.work[gid] : gid;
  if (active && p.status[pair] != WFA_ST_DONE) active = false;

  int plen = 0, tlen = 0;
  const uint32_t* Pw = nullptr; const uint32_t* Tw = nullptr;
  const uint8_t* q = nullptr; uint32_t nops = 0;
  bool fail = false;
  if (active) {
    const WfaSeqPair mp = p.meta[pair];
    plen = (int)mp.pattern_len; tlen = (int)mp.text_len;
    Pw = p.packed + ((p.raw ? mp.pattern_offset : mp.pattern_offset_packed) >> 2);
    Tw = p.packed + ((p.raw ? mp.text_offset : mp.text_offset_packed) >> 2);
    const unsigned long long off = p.cigar_off[pair];
    q = ((off & WFA_OPS_IN_ARENA) ? p.arena : p.ops) + (off & ~WFA_OPS_IN_ARENA);
    nops = p.cigar_len[pair];
    fail = nops == 0xFFFFFFFFu;
  }
  if constexpr (SEQ_LDS) stage_pairs(p, PPW, lane, active, plen, tlen, Pw, Tw, seq_lds);
  const uint8_t* q_lds = nullptr;
  if constexpr (OPS_LDS) {
    // long alignments (the wavefront kernel walked): the op lists -- thousands of operations, read four per load by ONE lane, a
    // round trip to L2 each time -- come into LDS too, after the sequences, copied by the whole wavefront
    uint32_t* const ops_l = seq_lds + (size_t)PPW * p.seq_lds_stride;
    const uint32_t lwords = ((p.ops_slot >> 2) + 1u) | 1u;
    const uint32_t my_words = (active && !fail) ? (nops + (uint32_t)(reinterpret_cast<uintptr_t>(q) & 3) + 3u) >> 2 : 0u;
    const uint32_t* const my_src = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(q) & ~(uintptr_t)3);
    using GlobalWords = const __attribute__((address_space(1))) uint32_t*;
    for (int j = 0; j < PPW; ++j) {
      const uint32_t nw = __builtin_amdgcn_readlane(my_words, j);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(reinterpret_cast<unsigned long long>(my_src) >> 32), j) << 32) |
                                                              (uint32_t)__builtin_amdgcn_readlane((int)reinterpret_cast<unsigned long long>(my_src), j));
      uint32_t* dst = ops_l + (size_t)j * lwords;
      for (uint32_t i = lane; i < nw; i += 64) dst[i] = ((GlobalWords)src)[i];
    }
    __syncthreads();
    // (a variable of its own: `q` stays a pointer to global memory for the compiler, this one a pointer to LDS)
    q_lds = reinterpret_cast<const uint8_t*>(ops_l + (size_t)lane * lwords) + (reinterpret_cast<uintptr_t>(q) & 3);
  }
  if (p.text_scratch) {
    // single replay: the text goes to this lane's slot of the scratch (sized by the same bound the host sizes the arenas
    // with: at most score / min(x, e) operations, each with a match run, item_chars characters per item); wfa_text_compact_kernel
    // moves it to its place in the dense arena
    const uint32_t bound = (active && !fail) ? (uint32_t)p.item_chars * (2u * ((uint32_t)p.score[pair] / (uint32_t)p.min_op_cost) + 1u) + 1u : 0u;
    const unsigned long long s_off = wave_alloc(p.scratch_top, bound, lane);
    if (active && !fail && s_off + bound > p.text_scratch_cap) fail = true;
    if (active) {
      uint32_t len = 0xFFFFFFFFu;
      if (!fail) {
        int cost = 0;
        if constexpr (OPS_LDS)
          len = p.raw ? replay<true, true>(q_lds, nops, Pw, Tw, plen, tlen, p.text_scratch + s_off, p.x, p.oe - p.e, p.e, &cost, bound, true)
                      : replay<false, true>(q_lds, nops, Pw, Tw, plen, tlen, p.text_scratch + s_off, p.x, p.oe - p.e, p.e, &cost, bound, true);
        else
          len = p.raw ? replay<true>(q, nops, Pw, Tw, plen, tlen, p.text_scratch + s_off, p.x, p.oe - p.e, p.e, &cost, bound, true)
                      : replay<false>(q, nops, Pw, Tw, plen, tlen, p.text_scratch + s_off, p.x, p.oe - p.e, p.e, &cost, bound, true);
        if (len != 0xFFFFFFFFu && len + 1u > bound) len = 0xFFFFFFFFu;
        if (len != 0xFFFFFFFFu && cost != p.score[pair]) {
          if (p.score_fix) p.score_fix[pair] = cost;
          else len = 0xFFFFFFFFu;
        }
      }
      p.cigar_off[pair] = s_off;
      p.cigar_len[pair] = len;
    }
    return;
  }
  uint32_t len = 0;
  if (active && !fail) {
    len = p.raw ? replay<true>(q, nops, Pw, Tw, plen, tlen, nullptr, 0, 0, 0, nullptr)
                : replay<false>(q, nops, Pw, Tw, plen, tlen, nullptr, 0, 0, 0, nullptr);
    if (len == 0xFFFFFFFFu) fail = true;
  }
  const uint32_t need_txt = (active && !fail) ? len + 1u : 0u;
  const unsigned long long txt_off = wave_alloc(p.text_top, need_txt, lane);
  if (active && !fail && txt_off + need_txt > p.text_cap) fail = true;
  if (active) {
    if (!fail) {
      int cost = 0;
      if (p.raw) replay<true>(q, nops, Pw, Tw, plen, tlen, p.text + txt_off, p.x, p.oe - p.e, p.e, &cost);
      else replay<false>(q, nops, Pw, Tw, plen, tlen, p.text + txt_off, p.x, p.oe - p.e, p.e, &cost);
      p.cigar_off[pair] = txt_off;
      p.cigar_len[pair] = len;
      // The text's own gap-affine cost.  Exact alignments: it equals the wavefront score (checked: a
      // difference would mean a broken trace).  Adaptive band: a path may open two gaps back to back
      // where the extension cell had left the band; printed, they are one gap and cost less -- the
      // reported score follows the alignment that is returned.
      if (cost != p.score[pair]) {
        if (p.score_fix) p.score_fix[pair] = cost;
        else p.cigar_len[pair] = 0xFFFFFFFFu;
      }
    } else {
      p.cigar_off[pair] = 0;
      p.cigar_len[pair] = 0xFFFFFFFFu;
    }
  }
}

// Texts from their scratch slots (cigar_off = scratch offset, cigar_len = length) to the dense arena: one wavefront per
// 64 alignments, space bought with one atomic per wavefront, every text copied by all 64 lanes.
// (four wavefronts per workgroup share ONE atomic: a returning atomic on the one bump counter costs ~11 ns whoever issues it, and
// one per wavefront was 0.17 of this kernel's 0.34 ms per 1M texts)
constexpr int COMPACT_WAVES = 4;
__global__ void __launch_bounds__(COMPACT_WAVES * 64) wfa_text_compact_kernel(const WfaTraceParams p) {
  __shared__ uint32_t wave_total[COMPACT_WAVES];
  __shared__ unsigned long long block_base;
  const uint32_t gid = blockIdx.x * (COMPACT_WAVES * 64) + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool active = gid < p.n_work;
  uint32_t pair = 0;
  if (active) pair = p.work ? p.work[gid] : gid;
  if (active && p.status[pair] != WFA_ST_DONE) active = false;
  uint32_t len = active ? p.cigar_len[pair] : 0xFFFFFFFFu;
  const unsigned long long src = active ? p.cigar_off[pair] : 0ull;
  const uint32_t need = len != 0xFFFFFFFFu ? len + 1u : 0u;
  const unsigned long long dst = block_alloc<COMPACT_WAVES>(p.text_top, need, wave_total, &block_base);
  if (need && dst + need > p.text_cap) len = 0xFFFFFFFFu;
  // Eight texts at a time, eight lanes each, 32-bit words (unaligned on both sides), up to eight words per lane loaded
  // before the first store: a text was copied byte-wise by the whole wavefront, one load-store round trip after the other.
  const int grp = lane >> 3, sub = lane & 7;
  for (int j0 = 0; j0 < 64; j0 += 8) {
    const int j = j0 + grp;
    const uint32_t nj = __shfl(len != 0xFFFFFFFFu ? len + 1u : 0u, j);
    const char* sj = p.text_scratch + shfl64(src, j);
    char* dj = p.text + shfl64(dst, j);
    const uint32_t nmax = __builtin_amdgcn_readfirstlane(max(max(__shfl(nj, 0 * 8), __shfl(nj, 1 * 8)), max(max(__shfl(nj, 2 * 8), __shfl(nj, 3 * 8)),
                                                        max(max(__shfl(nj, 4 * 8), __shfl(nj, 5 * 8)), max(__shfl(nj, 6 * 8), __shfl(nj, 7 * 8))))));
    for (uint32_t b0 = 0; b0 < nmax; b0 += 256u) {
      uint32_t w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t t = b0 + 4u * (uint32_t)sub + 32u * (uint32_t)u;
        w[u] = 0;
        if (t + 4u <= nj) __builtin_memcpy(&w[u], sj + t, 4);
        else if (t < nj) { for (uint32_t q = t; q < nj; ++q) w[u] |= (uint32_t)(uint8_t)sj[q] << (8u * (q - t)); }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t t = b0 + 4u * (uint32_t)sub + 32u * (uint32_t)u;
        if (t + 4u <= nj) __builtin_memcpy(dj + t, &w[u], 4);
        else if (t < nj) { for (uint32_t q = t; q < nj; ++q) dj[q] = (char)(w[u] >> (8u * (q - t))); }
      }
    }
  }
  if (active) {
    p.cigar_off[pair] = len != 0xFFFFFFFFu ? dst : 0ull;
    p.cigar_len[pair] = len;
  }
}


// ---- long alignments: one WAVEFRONT per alignment ----------------------------------------------------------------
// With one lane per alignment a batch of long reads (1024 x 30 kbp: 3000 operations each) keeps 16 wavefronts busy on
// a 1024-SIMD chip and every step of the walk and of the replay pays a full memory round trip (walk 4.3 ms + emit
// 10.8 ms per 1024 pairs of BASELINE configs[4]).  Here the wavefront
//   1. copies both packed sequences into LDS (coalesced), so the replay's LCPs never leave the CU;
//   2. walks the origin bytes through TILES: lane r fetches 16 origin bytes of score s-r around the current diagonal
//      (64 independent loads in flight instead of one), then the walk consumes the tile from LDS until the path leaves
//      it (a tile lasts ~20 operations: mismatches stay on their diagonal, gaps move one diagonal per base);
//   3. replays the operations (two passes: size, then text) -- sequential, but out of LDS.
// All lanes follow the same control flow; lane 0 does the stores.
template <bool RAW>
__global__ void __launch_bounds__(TRACE_THREADS) wfa_trace_wave_kernel(const WfaTraceParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t wlds[];
  const int lane = threadIdx.x & 63;
  // WALK ONLY (p.walk_only): the wavefront walks -- tiles, below -- and leaves the op list in the global scratch (slot w of
  // p.ops_slot bytes, no allocation); the replay is the lane-per-alignment kernel's, with the sequences of a few pairs per
  // wavefront staged in LDS (wfa_emit_kernel).  A replay is one serial chain per alignment: here every one of the 64 lanes
  // executed it (lane 0 stored), five such wavefronts per SIMD taking turns on the vector pipe; there, 8-16 alignments share a
  // wavefront and the pipe is nearly idle.  No sequences in this kernel's LDS then: 1 KB of tile + the op list.
  const int seq_cap = p.walk_only ? 0 : p.seq_words_cap;
  uint32_t* Pw = wlds;
  uint32_t* Tw = Pw + seq_cap;
  uint8_t* tile = reinterpret_cast<uint8_t*>(Tw + seq_cap);     // [64 rows][32 bytes] (16 used by the row-table layout, 32 by tiles)
  uint8_t* ops_lds = tile + 64 * TILE_LDS_W;                            // [ops_lds_bytes]
  char* text_lds = reinterpret_cast<char*>(ops_lds + p.ops_lds_bytes);  // [text_lds_bytes]
  constexpr int TILE_H = 64;
  for (uint32_t w = blockIdx.x; w < p.n_work; w += gridDim.x) {
    const uint32_t pair = __builtin_amdgcn_readfirstlane(p.work ? p.work[w] : w);
    if (p.status[pair] != WFA_ST_DONE) continue;
    const WfaSeqPair mp = p.meta[pair];
    const int plen = __builtin_amdgcn_readfirstlane((int)mp.pattern_len), tlen = __builtin_amdgcn_readfirstlane((int)mp.text_len);
    const int score = __builtin_amdgcn_readfirstlane(p.score[pair]);
    const int sh = RAW ? 2 : 4;
    const int pwords = ((plen + (1 << sh) - 1) >> sh) + 1, twords = ((tlen + (1 << sh) - 1) >> sh) + 1;
    if (!p.walk_only) {
      const uint32_t* gp = p.packed + ((RAW ? mp.pattern_offset : mp.pattern_offset_packed) >> 2);
      const uint32_t* gt = p.packed + ((RAW ? mp.text_offset : mp.text_offset_packed) >> 2);
      for (int i = lane; i < pwords; i += 64) Pw[i] = gp[i];
      for (int i = lane; i < twords; i += 64) Tw[i] = gt[i];
    }
    // the reversed op list (every operation costs at least min(x, e) >= 1: `score` bytes suffice): in LDS when it
    // fits -- the replay reads one op per step, a round trip to L2 each if they sat in global memory --, else in the
    // global scratch
    const uint32_t need_ops = ((uint32_t)score + 3u) & ~3u;
    bool fail = false;
    uint8_t* q_begin = ops_lds;
    if (need_ops > (uint32_t)p.ops_lds_bytes) {
      unsigned long long ops_off = (unsigned long long)w * p.ops_slot;       // (walk only: slot w, no allocation)
      if (!p.walk_only) {
        if (lane == 0) ops_off = atomicAdd(p.ops_top, (unsigned long long)need_ops);
        ops_off = shfl64(ops_off, 0);
      }
      fail = ops_off + need_ops > p.ops_cap;
      q_begin = p.ops + ops_off;
    }
    if (p.walk_only && need_ops > p.ops_slot) fail = true;
    uint8_t* const q_end = q_begin + need_ops;
    uint8_t* q = q_end;
    if (!fail) {
      const BtBlock bb = bt_block(p, pair);
      int k = tlen - plen, s = score, state = 0;       // state 0: M, 1: I, 2: D
      int s_top = -1, kbase = 0;
      const int tile_w = bb.tiled ? 32 : 16;
      while (s > 0) {
        if (s > s_top || s_top - s >= TILE_H || (unsigned)(k - kbase) >= (unsigned)tile_w) {
          // new tile: scores s .. s-63, diagonals k-8 .. k+7 (row-table layout) / the two 16-diagonal tile columns around k (tiles)
          s_top = s;
          __builtin_amdgcn_wave_barrier();
          fetch_bt_rows(p, bb, s - lane, k, kbase, tile + lane * TILE_LDS_W);
          __builtin_amdgcn_wave_barrier();
        }
        if (q == q_begin) { fail = true; break; }
        const uint32_t code = tile[(s_top - s) * TILE_LDS_W + (k - kbase)];
        uint8_t op;
        if (state == 0) {
          const uint32_t org = code & BT_M_MASK;
          if (org == BT_M_X) { op = OP_X | OP_EXT_AFTER; s -= p.x; }
          else if (org == BT_M_I) {
            op = OP_I | OP_EXT_AFTER; --k;
            if (code & BT_I_EXT) { s -= p.e; state = 1; } else { s -= p.oe; }
          } else if (org == BT_M_D) {
            op = OP_D | OP_EXT_AFTER; ++k;
            if (code & BT_D_EXT) { s -= p.e; state = 2; } else { s -= p.oe; }
          } else { fail = true; break; }
        } else if (state == 1) {
          op = OP_I; --k;
          if (code & BT_I_EXT) { s -= p.e; } else { s -= p.oe; state = 0; }
        } else {
          op = OP_D; ++k;
          if (code & BT_D_EXT) { s -= p.e; } else { s -= p.oe; state = 0; }
        }
        --q;
        if (lane == 0) *q = op;
      }
      if (s != 0 || state != 0 || k != 0) fail = true;
    }
    // the op list is read back by this same wavefront: make lane 0's stores visible to all lanes
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    const uint32_t nops = (uint32_t)(q_end - q);
    if (p.walk_only) {
      // the list goes to (or is already in) this work item's slot of the global scratch, where wfa_emit_kernel looks for it
      uint8_t* const g_end = p.ops + (size_t)w * p.ops_slot + need_ops;
      if (!fail && q_begin == ops_lds) { for (uint32_t i = lane; i < nops; i += 64) g_end[(long)i - (long)nops] = q[i]; }
      if (lane == 0) {
        p.cigar_off[pair] = fail ? 0ull : (unsigned long long)(g_end - nops - p.ops);
        p.cigar_len[pair] = fail ? 0xFFFFFFFFu : nops;
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    // One replay writes the text into LDS (every lane follows along, lane 0 stores); its length then buys the space
    // in the dense text arena and all 64 lanes copy it out.  A text longer than the LDS buffer is replayed once more,
    // straight into the arena.
    uint32_t len = 0;
    int cost = 0;
    if (!fail) {
      len = replay<RAW>(q, nops, Pw, Tw, plen, tlen, lane == 0 ? text_lds : nullptr, p.x, p.oe - p.e, p.e, &cost, (uint32_t)p.text_lds_bytes);
      if (len == 0xFFFFFFFFu) fail = true;
    }
    unsigned long long txt_off = 0;
    if (!fail) {
      if (lane == 0) txt_off = atomicAdd(p.text_top, (unsigned long long)len + 1ull);
      txt_off = shfl64(txt_off, 0);
      if (txt_off + len + 1 > p.text_cap) fail = true;
    }
    if (!fail) {
      if (len + 1u <= (uint32_t)p.text_lds_bytes) {
        __builtin_amdgcn_wave_barrier();
        char* dst = p.text + txt_off;
        for (uint32_t i = lane; i <= len; i += 64) dst[i] = text_lds[i];
      } else {
        replay<RAW>(q, nops, Pw, Tw, plen, tlen, lane == 0 ? p.text + txt_off : nullptr, p.x, p.oe - p.e, p.e, &cost);
      }
      if (lane == 0) {
        p.cigar_off[pair] = txt_off;
        p.cigar_len[pair] = len;
        if (cost != score) {
          if (p.score_fix) p.score_fix[pair] = cost;
          else p.cigar_len[pair] = 0xFFFFFFFFu;
        }
      }
    } else if (lane == 0) {
      p.cigar_off[pair] = 0;
      p.cigar_len[pair] = 0xFFFFFFFFu;
    }
    __builtin_amdgcn_wave_barrier();      // (the next alignment overwrites the staged sequences)
  }
}


// ---- alignments a little too long for the lane-per-alignment kernels: G alignments per wavefront -------------------
// One wavefront per alignment issues its walk and replay instructions for ONE alignment; with a hundred operations per
// alignment and hundreds of thousands of alignments (2 kbp reads) that is mostly launch and staging overhead per wave.  Here the 64 lanes form G groups of L = 64/G lanes, each group with its own alignment and its
// own share of LDS (sequences, tile of L rows x 16 origin bytes, op list, text): the L lanes of a group stage, fetch
// tiles and copy the text out together; the walk and the replay run on the group's first lane -- G of them per wave
// instruction.  Groups progress independently (divergent loops), everything a group shares goes through its LDS share.
template <bool RAW, int G>
__global__ void __launch_bounds__(TRACE_THREADS) wfa_trace_group_kernel(const WfaTraceParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t glds[];
  constexpr int L = 64 / G, TILE_H = L;       // one row per lane of the group (short alignments: taller tiles cost more than they save)
  const int lane = threadIdx.x & 63, grp = lane / L, sub = lane % L, lead = grp * L;
  const size_t share_words = ((size_t)2 * p.seq_words_cap * 4 + (size_t)TILE_H * TILE_LDS_W + (size_t)p.ops_lds_bytes + (size_t)p.text_lds_bytes + 15) / 16 * 4;
  uint32_t* Pw = glds + (size_t)grp * share_words;
  uint32_t* Tw = Pw + p.seq_words_cap;
  uint8_t* tile = reinterpret_cast<uint8_t*>(Tw + p.seq_words_cap);       // [L rows][32 bytes]
  uint8_t* ops_lds = tile + TILE_H * TILE_LDS_W;
  char* text_lds = reinterpret_cast<char*>(ops_lds + p.ops_lds_bytes);
  for (uint32_t base = blockIdx.x * G; base < p.n_work; base += gridDim.x * G) {
    const uint32_t w = base + grp;
    bool active = w < p.n_work;
    uint32_t pair = 0;
    if (active) pair = p.work ? p.work[w] : w;
    if (active && p.status[pair] != WFA_ST_DONE) active = false;
    int plen = 0, tlen = 0, score = 0, pwords = 0, twords = 0;
    if (active) {
      const WfaSeqPair mp = p.meta[pair];
      plen = (int)mp.pattern_len; tlen = (int)mp.text_len; score = p.score[pair];
      const int sh = RAW ? 2 : 4;
      pwords = ((plen + (1 << sh) - 1) >> sh) + 1; twords = ((tlen + (1 << sh) - 1) >> sh) + 1;
      const uint32_t* gp = p.packed + ((RAW ? mp.pattern_offset : mp.pattern_offset_packed) >> 2);
      const uint32_t* gt = p.packed + ((RAW ? mp.text_offset : mp.text_offset_packed) >> 2);
      for (int i = sub; i < pwords; i += L) Pw[i] = gp[i];
      for (int i = sub; i < twords; i += L) Tw[i] = gt[i];
    }
    const uint32_t need_ops = active ? (((uint32_t)score + 3u) & ~3u) : 0u;
    bool fail = false;
    uint8_t* q_begin = ops_lds;
    if (need_ops > (uint32_t)p.ops_lds_bytes) {
      unsigned long long ops_off = 0;
      if (sub == 0) ops_off = atomicAdd(p.ops_top, (unsigned long long)need_ops);
      ops_off = shfl64(ops_off, lead);
      fail = ops_off + need_ops > p.ops_cap;
      q_begin = p.ops + ops_off;
    }
    uint8_t* const q_end = q_begin + need_ops;
    uint8_t* q = q_end;
    if (active && !fail) {
      const BtBlock bb = bt_block(p, pair);
      int k = tlen - plen, s = score, state = 0;       // state 0: M, 1: I, 2: D
      int s_top = -1, kbase = 0;
      const int tile_w = bb.tiled ? 32 : 16;
      while (s > 0) {
        if (s > s_top || s_top - s >= TILE_H || (unsigned)(k - kbase) >= (unsigned)tile_w) {
          // new tile for this group: scores s .. s-L+1, diagonals around k (lane `sub` fetches row s - sub)
          s_top = s;
          fetch_bt_rows(p, bb, s - sub, k, kbase, tile + sub * TILE_LDS_W);
          __builtin_amdgcn_wave_barrier();
        }
        if (q == q_begin) { fail = true; break; }
        const uint32_t code = tile[(s_top - s) * TILE_LDS_W + (k - kbase)];
        uint8_t op;
        if (state == 0) {
          const uint32_t org = code & BT_M_MASK;
          if (org == BT_M_X) { op = OP_X | OP_EXT_AFTER; s -= p.x; }
          else if (org == BT_M_I) {
            op = OP_I | OP_EXT_AFTER; --k;
            if (code & BT_I_EXT) { s -= p.e; state = 1; } else { s -= p.oe; }
          } else if (org == BT_M_D) {
            op = OP_D | OP_EXT_AFTER; ++k;
            if (code & BT_D_EXT) { s -= p.e; state = 2; } else { s -= p.oe; }
          } else { fail = true; break; }
        } else if (state == 1) {
          op = OP_I; --k;
          if (code & BT_I_EXT) { s -= p.e; } else { s -= p.oe; state = 0; }
        } else {
          op = OP_D; ++k;
          if (code & BT_D_EXT) { s -= p.e; } else { s -= p.oe; state = 0; }
        }
        --q;
        if (sub == 0) *q = op;
      }
      if (s != 0 || state != 0 || k != 0) fail = true;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    const uint32_t nops = (uint32_t)(q_end - q);
    // replay on the group's first lane: once into the LDS text buffer (longer texts: a second time into the arena)
    uint32_t len = 0;
    int cost = 0;
    if (active && !fail && sub == 0) {
      len = replay<RAW>(q, nops, Pw, Tw, plen, tlen, text_lds, p.x, p.oe - p.e, p.e, &cost, (uint32_t)p.text_lds_bytes);
    }
    len = __shfl(len, lead);
    if (active && !fail && len == 0xFFFFFFFFu) fail = true;
    // (one returning atomic per wavefront for the texts of its G alignments)
    unsigned long long txt_off = wave_alloc(p.text_top, (active && !fail && sub == 0) ? len + 1u : 0u, lane);
    txt_off = shfl64(txt_off, lead);
    if (active && !fail && txt_off + len + 1 > p.text_cap) fail = true;
    if (active && !fail) {
      if (len + 1u <= (uint32_t)p.text_lds_bytes) {
        __builtin_amdgcn_wave_barrier();
        char* dst = p.text + txt_off;
        for (uint32_t i = sub; i <= len; i += L) dst[i] = text_lds[i];
      } else if (sub == 0) {
        replay<RAW>(q, nops, Pw, Tw, plen, tlen, p.text + txt_off, p.x, p.oe - p.e, p.e, &cost);
      }
    }
    if (active && sub == 0) {
      if (!fail) {
        p.cigar_off[pair] = txt_off;
        p.cigar_len[pair] = len;
        if (cost != score) {
          if (p.score_fix) p.score_fix[pair] = cost;
          else p.cigar_len[pair] = 0xFFFFFFFFu;
        }
      } else {
        p.cigar_off[pair] = 0;
        p.cigar_len[pair] = 0xFFFFFFFFu;
      }
    }
    __builtin_amdgcn_wave_barrier();      // (the next alignments overwrite the staged sequences)
  }
}

// The opt-in for dynamic LDS beyond 64 KiB is sticky per kernel and device: one driver call when the request grows.
template <typename K>
void allow_lds(K kernel, size_t lds, size_t (&allowed)[16]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > allowed[dev & 15]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    allowed[dev & 15] = lds;
  }
}

template <bool RAW, int G>
void launch_group(const WfaTraceParams& p, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  const size_t share = ((size_t)2 * p.seq_words_cap * 4 + (size_t)(64 / G) * TILE_LDS_W + (size_t)p.ops_lds_bytes + (size_t)p.text_lds_bytes + 15) / 16 * 16;
  const size_t lds = share * G;
  const uint32_t blocks = (p.n_work + G - 1) / G;
  const uint32_t grid = blocks < 8192u ? blocks : 8192u;
  static thread_local size_t allowed[16] = {0};
  allow_lds(wfa_trace_group_kernel<RAW, G>, lds, allowed);
  wfa_launch_timed(wfa_trace_group_kernel<RAW, G>, dim3(grid), dim3(TRACE_THREADS), lds, stream, ev0, ev1, p);
}

template <bool RAW>
void launch_wave(const WfaTraceParams& p, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  const size_t lds = p.walk_only ? (size_t)64 * TILE_LDS_W + (size_t)p.ops_lds_bytes
                                 : (size_t)2 * p.seq_words_cap * 4 + 64 * TILE_LDS_W + (size_t)p.ops_lds_bytes + (size_t)p.text_lds_bytes;
  const uint32_t grid = p.n_work < 8192u ? p.n_work : 8192u;
  static thread_local size_t allowed[16] = {0};
  allow_lds(wfa_trace_wave_kernel<RAW>, lds, allowed);
  wfa_launch_timed(wfa_trace_wave_kernel<RAW>, dim3(grid), dim3(TRACE_THREADS), lds, stream, ev0, ev1, p);
}

}  // namespace

// ev0 / ev1: receive the start of the first and the end of the last kernel launched here (a call that launches nothing --
// n_work == 0 -- leaves them alone: returns false)
bool wfa_launch_trace(const WfaTraceParams& p, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  if (p.n_work == 0) return false;
  if (p.wave_kernel && p.group > 1) {
    switch (p.group) {
      case 8: if (p.raw) launch_group<true, 8>(p, stream, ev0, ev1); else launch_group<false, 8>(p, stream, ev0, ev1); break;
      case 4: if (p.raw) launch_group<true, 4>(p, stream, ev0, ev1); else launch_group<false, 4>(p, stream, ev0, ev1); break;
      default: if (p.raw) launch_group<true, 2>(p, stream, ev0, ev1); else launch_group<false, 2>(p, stream, ev0, ev1); break;
    }
    return true;
  }
  if (p.wave_kernel && !p.walk_only) {
    if (p.raw) launch_wave<true>(p, stream, ev0, ev1); else launch_wave<false>(p, stream, ev0, ev1);
    return true;
  }
  if (p.wave_kernel) {
    // long alignments: the wavefront kernel walks, the lane kernel replays (a few pairs per wavefront staged in LDS)
    if (p.raw) launch_wave<true>(p, stream, ev0, (hipEvent_t) nullptr); else launch_wave<false>(p, stream, ev0, (hipEvent_t) nullptr);
    const size_t lds_e = (size_t)p.emit_pairs * ((size_t)p.seq_lds_stride * 4 + (size_t)((((p.ops_slot >> 2) + 1u) | 1u) * 4u));
    static thread_local size_t allowed_e[16] = {0};
    allow_lds(wfa_emit_kernel<true, true>, lds_e, allowed_e);
    const dim3 grid_e((p.n_work + (uint32_t)p.emit_pairs - 1) / (uint32_t)p.emit_pairs);
    wfa_launch_timed(wfa_emit_kernel<true, true>, grid_e, dim3(TRACE_THREADS), lds_e, stream, (hipEvent_t) nullptr, p.text_scratch ? (hipEvent_t) nullptr : ev1, p);
    if (p.text_scratch)
      wfa_launch_timed(wfa_text_compact_kernel, dim3((p.n_work + COMPACT_WAVES * 64 - 1) / (COMPACT_WAVES * 64)), dim3(COMPACT_WAVES * 64), 0, stream, (hipEvent_t) nullptr, ev1, p);
    return true;
  }
  const dim3 grid((p.n_work + TRACE_THREADS - 1) / TRACE_THREADS), block(TRACE_THREADS);
  if (p.lane_fused) {
    // short alignments: walk + both replays in one kernel, op lists in LDS (sequences, then ops_lds_bytes per lane, then 72 bytes of
    // row-table cache per lane)
    const size_t lds = (size_t)LANE_WAVES * lane_kernel_wave_bytes(p.emit_pairs, p.seq_lds_stride, p.ops_lds_bytes);
    const uint32_t per_block = (uint32_t)LANE_WAVES * (uint32_t)p.emit_pairs;
    const dim3 grid_f((p.n_work + per_block - 1) / per_block), block_f(LANE_WAVES * 64);
    static thread_local size_t allowed[2][16] = {{0}};
    if (p.raw) { allow_lds(wfa_trace_lane_kernel<true>, lds, allowed[1]); wfa_launch_timed(wfa_trace_lane_kernel<true>, grid_f, block_f, lds, stream, ev0, ev1, p); }
    else { allow_lds(wfa_trace_lane_kernel<false>, lds, allowed[0]); wfa_launch_timed(wfa_trace_lane_kernel<false>, grid_f, block_f, lds, stream, ev0, ev1, p); }
    return true;
  }
  uint32_t walk_grid = (p.n_work + WALK_WAVES * 64 - 1) / (WALK_WAVES * 64);
  if (p.walk_grid_cap > 0 && walk_grid > (uint32_t)p.walk_grid_cap) walk_grid = (uint32_t)p.walk_grid_cap;
  hipEvent_t ev_first = ev0;      // (start of whichever kernel comes first)
  if (!p.skip_walk) { wfa_launch_timed(wfa_walk_kernel, dim3(walk_grid), dim3(WALK_WAVES * 64), 0, stream, ev0, (hipEvent_t) nullptr, p); ev_first = nullptr; }
  if (p.seq_lds_stride == 0) {      // (sequences too long to stage 64 pairs, or tuning.trace_mode 1: 8-word LDS windows)
    wfa_launch_timed(wfa_emit_win_kernel, grid, block, 0, stream, ev_first, ev1, p);
    return true;
  }
  const size_t lds = (size_t)p.emit_pairs * p.seq_lds_stride * 4;
  static thread_local size_t allowed[16] = {0};
  allow_lds(wfa_emit_kernel<true, false>, lds, allowed);
  const dim3 grid_e((p.n_work + (uint32_t)p.emit_pairs - 1) / (uint32_t)p.emit_pairs);
  wfa_launch_timed(wfa_emit_kernel<true, false>, grid_e, block, lds, stream, ev_first, p.text_scratch ? (hipEvent_t) nullptr : ev1, p);
  if (p.text_scratch)
    wfa_launch_timed(wfa_text_compact_kernel, dim3((p.n_work + COMPACT_WAVES * 64 - 1) / (COMPACT_WAVES * 64)), dim3(COMPACT_WAVES * 64), 0, stream, (hipEvent_t) nullptr, ev1, p);
  return true;
}

// Loads this translation unit's code object on the current device (the runtime loads a code object at the first launch of
// any of its kernels: 5-25 ms each): launch_alignments* call it while a cold call waits for its first upload.
namespace { __global__ void k_prime_trace() {} }
void wfa_prime_trace(hipStream_t stream) { hipLaunchKernelGGL(k_prime_trace, dim3(1), dim3(64), 0, stream); }
