// 2-bit sequence packing for gfx950.
//
// Replaces the reference's compact_sequences kernel
// (lib/kernels/sequence_packing_kernel.cu:28-116) and its launcher
// (lib/sequence_packing.cu:96-116).  Same code assignment, (c & 6) >> 1, so
// A=0 C=1 T=2 G=3; different word layout (little-endian inside the 32-bit
// word, see wfa_device.h) and a different work decomposition: one 64-lane
// wavefront per pair, one lane per output word of each sequence, so every lane
// reads 16 contiguous ASCII bytes as four aligned dwords and writes one dword --
// both sides fully coalesced -- instead of the reference's byte-granular
// scattered stores.
//
// The kernel also reports, per sequence, whether any byte is outside
// {A,C,G,T}; such pairs must not go through the 2-bit path because WFA2
// (the ground truth) compares raw bytes.
#include "wfa_device.h"
#include "pack_device.h"

namespace {

using namespace wfa_pack;

constexpr int PACK_WAVES = 4;

// One wavefront per PAIR: the record is read once and the loads of both sequences are in flight together (one wavefront
// per sequence, with the last, partial word handled by dependent conditional loads, ran at 2.8 TB/s).  SHORT sequences: a group
// of L = 32, 16 or 8 lanes per pair, 64 / L pairs per wavefront -- a 150 bp read is ten output words, so with one pair per
// wavefront five lanes of six sat idle (BASELINE configs[1]: 46.9 -> 16 us per 100k pairs; it was a fifth of the resident step).
template <int L>
__global__ void __launch_bounds__(PACK_WAVES * 64)
wfa_pack_kernel(const char* __restrict__ ascii, const WfaSeqPair* __restrict__ meta,
                uint32_t n_pairs, uint32_t* __restrict__ packed, uint8_t* __restrict__ flags,
                uint32_t* __restrict__ status, unsigned long long* __restrict__ n_raw) {
  constexpr uint32_t G = 64 / L;
  const int lane = threadIdx.x & 63, j = lane % L, grp = lane / L;
  const uint32_t pair = (blockIdx.x * PACK_WAVES + (threadIdx.x >> 6)) * G + (uint32_t)grp;
  const bool have = pair < n_pairs;       // (whole groups: the ballots below need every lane of the wavefront)
  const WfaSeqPair m = meta[have ? pair : n_pairs - 1u];
  const uint32_t plen = have ? m.pattern_len : 0u, tlen = have ? m.text_len : 0u;
  // (an empty sequence may sit at the very end of the buffer: its -- fully masked -- loads go to the record array instead)
  const uint32_t* __restrict__ psrc = plen ? reinterpret_cast<const uint32_t*>(ascii + m.pattern_offset) : reinterpret_cast<const uint32_t*>(meta);
  const uint32_t* __restrict__ tsrc = tlen ? reinterpret_cast<const uint32_t*>(ascii + m.text_offset) : reinterpret_cast<const uint32_t*>(meta);
  uint32_t* __restrict__ pdst = packed + (m.pattern_offset_packed >> 2);
  uint32_t* __restrict__ tdst = packed + (m.text_offset_packed >> 2);
  const uint32_t pwords = (plen + 15u) >> 4, twords = (tlen + 15u) >> 4;   // + one spare word each, zeroed
  uint32_t pbad = 0, tbad = 0;
  if (have) {
    for (uint32_t w = j; w <= max(pwords, twords); w += L) {
      const PackWord pw = load_word(psrc, plen, min(w, pwords)), tw = load_word(tsrc, tlen, min(w, twords));
      const uint32_t po = pack_word(pw, plen, w, pbad), to = pack_word(tw, tlen, w, tbad);
      if (w <= pwords) pdst[w] = po;
      if (w <= twords) tdst[w] = to;
    }
  }
  const unsigned long long grp_mask = (L == 64) ? ~0ull : (((1ull << L) - 1ull) << (grp * L));
  const unsigned long long any_pbad = __ballot(pbad != 0) & grp_mask, any_tbad = __ballot(tbad != 0) & grp_mask;
  if (have && j == 0) { flags[2u * pair] = any_pbad ? 1 : 0; flags[2u * pair + 1u] = any_tbad ? 1 : 0; }
  // (status != nullptr -- the align call's own packing: flagged pairs never enter the 2-bit tiers, and their number lets the host
  // skip the byte-compare class when it is empty.  A kernel of its own did this: 4 us + the gap of a launch in a 175 us step of
  // 100k short reads; clean batches pay no atomic.)
  // (every pair's status is written here -- PENDING for the clean ones: the align call needs no memset of the status array in front
  // of this kernel, one launch fewer per step)
  if (status) {
    const bool raw = have && j == 0 && (any_pbad | any_tbad) != 0ull;
    if (have && j == 0) status[pair] = raw ? WFA_ST_ALPHABET : WFA_ST_PENDING;
    const unsigned long long bal = __ballot(raw);
    if (bal && lane == __builtin_ctzll(bal)) atomicAdd(n_raw, (unsigned long long)__builtin_popcountll(bal));
  }
}

}  // namespace

// max_seq_len: the longest sequence of the batch (0: unknown) -- picks the lanes per pair
void wfa_launch_pack(const char* d_ascii, const WfaSeqPair* d_meta, uint32_t n_pairs,
                     uint32_t* d_packed, uint8_t* d_flags, hipStream_t stream, uint32_t max_seq_len, hipEvent_t ev0, hipEvent_t ev1,
                     uint32_t* d_status, unsigned long long* d_n_raw) {
  if (n_pairs == 0) return;
  const uint32_t words = max_seq_len ? (max_seq_len + 15u) / 16u + 1u : 1u << 20;
  const uint32_t lanes = words <= 8u ? 8u : (words <= 16u ? 16u : (words <= 32u ? 32u : 64u));
  const uint32_t per_block = PACK_WAVES * (64u / lanes);
  const dim3 grid((n_pairs + per_block - 1) / per_block), block(PACK_WAVES * 64);
  switch (lanes) {
    case 8: wfa_launch_timed(wfa_pack_kernel<8>, grid, block, 0, stream, ev0, ev1, d_ascii, d_meta, n_pairs, d_packed, d_flags, d_status, d_n_raw); break;
    case 16: wfa_launch_timed(wfa_pack_kernel<16>, grid, block, 0, stream, ev0, ev1, d_ascii, d_meta, n_pairs, d_packed, d_flags, d_status, d_n_raw); break;
    case 32: wfa_launch_timed(wfa_pack_kernel<32>, grid, block, 0, stream, ev0, ev1, d_ascii, d_meta, n_pairs, d_packed, d_flags, d_status, d_n_raw); break;
    default: wfa_launch_timed(wfa_pack_kernel<64>, grid, block, 0, stream, ev0, ev1, d_ascii, d_meta, n_pairs, d_packed, d_flags, d_status, d_n_raw); break;
  }
}

// Loads this translation unit's code object on the current device (the runtime loads a code object at the first launch of
// any of its kernels: 5-25 ms each): launch_alignments* call it while a cold call waits for its first upload.
namespace { __global__ void k_prime_pack() {} }
void wfa_prime_pack(hipStream_t stream) { hipLaunchKernelGGL(k_prime_pack, dim3(1), dim3(64), 0, stream); }
