// 2-bit sequence packing for gfx950.
//
// Replaces the reference's compact_sequences kernel
// (lib/kernels/sequence_packing_kernel.cu:28-116) and its launcher
// (lib/sequence_packing.cu:96-116).  Same code assignment, (c & 6) >> 1, so
// A=0 C=1 T=2 G=3; different word layout (little-endian inside the 32-bit
// word, see wfa_device.h) and a different work decomposition: one 64-lane
// wavefront per sequence, one lane per output word, so every lane reads 16
// contiguous ASCII bytes as four aligned dwords and writes one dword --
// both sides fully coalesced -- instead of the reference's byte-granular
// scattered stores.
//
// The kernel also reports, per sequence, whether any byte is outside
// {A,C,G,T}; such pairs must not go through the 2-bit path because WFA2
// (the ground truth) compares raw bytes.
#include "wfa_device.h"

namespace {

__device__ __forceinline__ uint32_t code4(uint32_t w) {
  // 2-bit code of each of four ASCII bytes, in place: (c & 6) >> 1
  return (w >> 1) & 0x03030303u;
}

__device__ __forceinline__ uint32_t pack4(uint32_t codes) {
  // four byte-wide codes -> 8 bits, first byte in the low bits
  uint32_t t = (codes | (codes >> 6)) & 0x000F000Fu;
  t = (t | (t >> 12)) & 0xFFu;
  return t;
}

__device__ __forceinline__ bool bad4(uint32_t w, uint32_t codes) {
  // true if any of the four bytes is not one of A C G T: v_perm_b32 maps every code back to its
  // letter (byte table "ACTG"), which must reproduce the input
  return __builtin_amdgcn_perm(0x47544341u, 0x47544341u, codes) != w;
}

constexpr int PACK_WAVES = 4;

__global__ void __launch_bounds__(PACK_WAVES * 64)
wfa_pack_kernel(const char* __restrict__ ascii, const WfaSeqPair* __restrict__ meta,
                uint32_t n_pairs, uint32_t* __restrict__ packed, uint8_t* __restrict__ flags) {
  const uint32_t seq = blockIdx.x * PACK_WAVES + (threadIdx.x >> 6);  // 2*pair + {0: pattern, 1: text}
  const int lane = threadIdx.x & 63;
  if (seq >= 2u * n_pairs) return;
  const uint32_t pair = seq >> 1;
  const bool is_text = seq & 1u;
  const WfaSeqPair m = meta[pair];
  const uint32_t len = is_text ? m.text_len : m.pattern_len;
  const size_t src_off = is_text ? m.text_offset : m.pattern_offset;
  const size_t dst_off = is_text ? m.text_offset_packed : m.pattern_offset_packed;
  const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(ascii + src_off);
  uint32_t* __restrict__ dst = packed + (dst_off >> 2);
  const uint32_t n_words = (len + 15u) >> 4;   // + one spare word, zeroed
  uint32_t bad = 0;
  for (uint32_t w = lane; w <= n_words; w += 64) {
    uint32_t out = 0;
    if (w < n_words) {
      const uint32_t base = w << 4;             // first base of this word
      if (base + 16u <= len) {
        // all 16 bases present: four unconditional dword loads (one 16-byte access)
        const uint32_t* q4 = src + (base >> 2);
        const uint32_t a0 = q4[0], a1 = q4[1], a2 = q4[2], a3 = q4[3];
        const uint32_t c0 = code4(a0), c1 = code4(a1), c2 = code4(a2), c3 = code4(a3);
        bad |= (bad4(a0, c0) || bad4(a1, c1) || bad4(a2, c2) || bad4(a3, c3)) ? 1u : 0u;
        out = pack4(c0) | (pack4(c1) << 8) | (pack4(c2) << 16) | (pack4(c3) << 24);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t b = base + 4u * q;
          if (b < len) {
            uint32_t a = src[(base >> 2) + q];
            const uint32_t nvalid = len - b;       // >= 1
            if (nvalid < 4) {
              // bytes past the end are the buffer's NUL padding or the next
              // sequence: neutralise them with 'A' (code 0)
              const uint32_t keep = (1u << (8 * nvalid)) - 1u;
              a = (a & keep) | (0x41414141u & ~keep);
            }
            const uint32_t c4 = code4(a);
            bad |= bad4(a, c4) ? 1u : 0u;
            out |= pack4(c4) << (8 * q);
          }
        }
      }
    }
    dst[w] = out;
  }
  const unsigned long long any_bad = __ballot(bad != 0);
  if (lane == 0) flags[seq] = any_bad ? 1 : 0;
}

}  // namespace

void wfa_launch_pack(const char* d_ascii, const WfaSeqPair* d_meta, uint32_t n_pairs,
                     uint32_t* d_packed, uint8_t* d_flags, hipStream_t stream) {
  if (n_pairs == 0) return;
  const uint32_t n_seq = 2u * n_pairs;
  const uint32_t grid = (n_seq + PACK_WAVES - 1) / PACK_WAVES;
  hipLaunchKernelGGL(wfa_pack_kernel, dim3(grid), dim3(PACK_WAVES * 64), 0, stream,
                     d_ascii, d_meta, n_pairs, d_packed, d_flags);
}
