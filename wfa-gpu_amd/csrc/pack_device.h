// 2-bit packing of ASCII sequences on the device: the word arithmetic shared by wfa_pack_kernel (pack_kernel.hip) and by the
// wavefront kernels when they pack while staging (align_kernel.hip, WfaAlignParams::ascii).  Code assignment (c & 6) >> 1 --
// A=0 C=1 T=2 G=3, lib/kernels/sequence_packing_kernel.cu:28-116 of the reference --, little-endian inside the 32-bit word.
#ifndef WFA_PACK_DEVICE_H
#define WFA_PACK_DEVICE_H

#include "wfa_device.h"

namespace wfa_pack {

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

// One word (16 bases) of one sequence, branch-free: the four dwords are always loaded (indices clamped to the last dword
// the sequence touches, so nothing beyond it is ever read) and the bytes past the end -- the buffer's NUL padding or
// the next sequence -- are replaced by 'A' (code 0) with a mask; the spare word after the last one comes out as 0.
struct PackWord { uint32_t a0, a1, a2, a3; };
__device__ __forceinline__ PackWord load_word(const uint32_t* __restrict__ src, uint32_t len, uint32_t w) {
  const uint32_t last = len ? (len - 1u) >> 2 : 0u;
  const uint32_t q = w << 2;
  return PackWord{src[min(q, last)], src[min(q + 1u, last)], src[min(q + 2u, last)], src[min(q + 3u, last)]};
}
__device__ __forceinline__ uint32_t keep_valid(uint32_t a, uint32_t len, uint32_t first) {
  // bytes [first, first + 4) of the sequence: those at or beyond len become 'A'
  const uint32_t nvalid = first < len ? min(len - first, 4u) : 0u;
  const uint32_t keep = nvalid >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nvalid)) - 1u);
  return (a & keep) | (0x41414141u & ~keep);
}
__device__ __forceinline__ uint32_t pack_word(const PackWord& p, uint32_t len, uint32_t w, uint32_t& bad) {
  const uint32_t base = w << 4;
  const uint32_t a0 = keep_valid(p.a0, len, base), a1 = keep_valid(p.a1, len, base + 4u), a2 = keep_valid(p.a2, len, base + 8u),
                 a3 = keep_valid(p.a3, len, base + 12u);
  const uint32_t c0 = code4(a0), c1 = code4(a1), c2 = code4(a2), c3 = code4(a3);
  bad |= (bad4(a0, c0) || bad4(a1, c1) || bad4(a2, c2) || bad4(a3, c3)) ? 1u : 0u;
  return pack4(c0) | (pack4(c1) << 8) | (pack4(c2) << 16) | (pack4(c3) << 24);
}

}  // namespace wfa_pack

#endif
