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

// One word (16 bases) of one sequence, branch-free: the four dwords are always loaded (indices clamped to the last dword
// the sequence touches, so nothing beyond it is ever read) and the bytes past the end -- the buffer's NUL padding or
// the next sequence -- are replaced by 'A' (code 0) with a mask; the spare word after the last one comes out as 0.
struct PackWord { uint32_t a0, a1, a2, a3; };
__device__ __forceinline__ PackWord load_word(const uint32_t* __restrict__ src, uint32_t len, uint32_t w) {
  const uint32_t last = len ? (len - 1u) >> 2 : 0u;
  const uint32_t q = w << 2;
  return PackWord{src[min(q, last)], src[min(q + 1u, last)], src[min(q + 2u, last)], src[min(q + 3u, last)]};
}
// One packed word from its sixteen bytes.  Bytes at or beyond `len` -- the buffer's NUL padding or the next sequence -- count as 'A'
// (code 0) and are not looked at by the alphabet check.  ~40 vector instructions (the first version: ~85, which the wavefront
// kernels felt once they packed while staging): the validity of the sixteen bytes as two 64-bit byte masks, applied with one
// v_bfi per dword; the four 2-bit codes of a dword gathered into a byte by ONE multiplication (the codes sit 8 bits apart, the
// factor 2^24 + 2^18 + 2^12 + 2^6 moves them 6 bits apart into the top byte, no two terms collide); the alphabet check as an XOR
// against the letters the codes stand for, accumulated over the four dwords.
__device__ __forceinline__ uint32_t pack_word(const PackWord& p, uint32_t len, uint32_t w, uint32_t& bad) {
  const int nvalid = (int)len - (int)(w << 4);
  const int n_lo = min(max(nvalid, 0), 8), n_hi = min(max(nvalid - 8, 0), 8);
  const unsigned long long m_lo = n_lo >= 8 ? ~0ull : ((1ull << (8 * n_lo)) - 1ull);
  const unsigned long long m_hi = n_hi >= 8 ? ~0ull : ((1ull << (8 * n_hi)) - 1ull);
  auto keep = [](uint32_t mask, uint32_t a) { return (a & mask) | (0x41414141u & ~mask); };      // (v_bfi_b32)
  const uint32_t a0 = keep((uint32_t)m_lo, p.a0), a1 = keep((uint32_t)(m_lo >> 32), p.a1);
  const uint32_t a2 = keep((uint32_t)m_hi, p.a2), a3 = keep((uint32_t)(m_hi >> 32), p.a3);
  const uint32_t c0 = code4(a0), c1 = code4(a1), c2 = code4(a2), c3 = code4(a3);
  // every code mapped back to its letter (byte table "ACTG") must reproduce the input
  const uint32_t x = (__builtin_amdgcn_perm(0x47544341u, 0x47544341u, c0) ^ a0) | (__builtin_amdgcn_perm(0x47544341u, 0x47544341u, c1) ^ a1) |
                     (__builtin_amdgcn_perm(0x47544341u, 0x47544341u, c2) ^ a2) | (__builtin_amdgcn_perm(0x47544341u, 0x47544341u, c3) ^ a3);
  bad |= x != 0u ? 1u : 0u;
  constexpr uint32_t GATHER = (1u << 24) | (1u << 18) | (1u << 12) | (1u << 6);
  const uint32_t q0 = c0 * GATHER, q1 = c1 * GATHER, q2 = c2 * GATHER, q3 = c3 * GATHER;      // the packed byte of each dword: its top byte
  // bytes 3 of q0, q1 -> low half; of q2, q3 -> high half (v_perm_b32: selectors 0-3 = bytes of the second operand, 4-7 of the first)
  const uint32_t lo16 = __builtin_amdgcn_perm(q1, q0, 0x0c0c0703u), hi16 = __builtin_amdgcn_perm(q3, q2, 0x07030c0cu);
  return lo16 | hi16;
}

}  // namespace wfa_pack

#endif
