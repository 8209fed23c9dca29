/* 2-bit packing of one sequence on the host: the same words csrc/pack_kernel.hip writes on the device (code (c & 6) >> 1,
 * first base in the low bits of a little-endian 32-bit word, bytes past the end count as 'A', ceil(len / 16) words plus
 * one zero word).  launch_alignments* use it to send a quarter of the bytes over PCIe when the host has the cores
 * (csrc/wfa_launch.hip); the device kernel stays the path of resident batches and of every batch that holds a byte
 * outside ACGT (those pairs need their ASCII on the device: WFA2 compares raw bytes).
 *
 * Reference: lib/kernels/sequence_packing_kernel.cu:28-116 (the packing itself), lib/align.cu:224-236 (where it runs). */
#include "../../include/wfa_gpu_device.h"

#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

/* scalar: 8 bases per step, SWAR */
static inline uint32_t pack8_swar(uint64_t x) {
    uint64_t c = (x >> 1) & 0x0303030303030303ull;
    c = (c | (c >> 6)) & 0x000F000F000F000Full;
    c = (c | (c >> 12)) & 0x000000FF000000FFull;
    c = (c | (c >> 24)) & 0xFFFFull;
    return (uint32_t)c;
}

static inline int bad8_swar(uint64_t x) {
    /* letter of every code, from the byte table "ACTG" (0x41 0x43 0x54 0x47): must reproduce the input */
    const uint64_t c = (x >> 1) & 0x0303030303030303ull;
    const uint64_t b0 = c & 0x0101010101010101ull, b1 = (c >> 1) & 0x0101010101010101ull;
    /* A=0x41; C: +2; T: +0x13; G: +6  ->  0x41 + 2*b0 + 0x13*b1 - 0x0F*(b0&b1) */
    const uint64_t both = b0 & b1;
    const uint64_t letter = 0x4141414141414141ull + 2 * b0 + 0x13 * b1 - 0x0F * both;
    return letter != x;
}

static int pack_tail(const unsigned char* src, uint32_t from, uint32_t len, uint32_t* dst) {
    /* words from base `from` (a multiple of 16) to the end, then the zero word */
    int bad = 0;
    uint32_t w = from >> 4;
    for (uint32_t i = from; i < len; i += 16, ++w) {
        uint32_t word = 0;
        const uint32_t n = len - i < 16 ? len - i : 16;
        for (uint32_t j = 0; j < n; ++j) {
            const unsigned char ch = src[i + j];
            bad |= !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
            word |= (uint32_t)((ch >> 1) & 3u) << (2 * j);
        }
        dst[w] = word;
    }
    dst[w] = 0;
    return bad;
}

/* words from base `from` (a multiple of 16) on */
static int pack_scalar(const unsigned char* src, uint32_t from, uint32_t len, uint32_t* dst) {
    int bad = 0;
    const uint32_t full = len & ~15u;
    for (uint32_t i = from; i < full; i += 16) {
        uint64_t a, b;
        memcpy(&a, src + i, 8); memcpy(&b, src + i + 8, 8);
        bad |= bad8_swar(a) | bad8_swar(b);
        dst[i >> 4] = pack8_swar(a) | (pack8_swar(b) << 16);
    }
    return bad | pack_tail(src, full > from ? full : from, len, dst);
}

#if defined(__x86_64__)
/* `avail` >= len: bytes that may be READ from src (the caller's buffer goes on that far): lets the last, partial 32 bytes
 * go through the vector path too -- short reads are mostly tail */
__attribute__((target("avx2")))
static int pack_avx2(const unsigned char* src, uint32_t len, size_t avail, uint32_t* dst) {
    const __m256i table = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                           'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i three = _mm256_set1_epi8(3);
    __m256i ok = _mm256_set1_epi8(-1);
    const uint32_t full = len & ~31u;
    /* codes -> bits: c0 + 4 c1 per 16-bit lane (vpmaddubsw), then n0 + 16 n1 per 32-bit lane (vpmaddwd): one byte of
     * four bases in every dword, gathered by a byte shuffle */
    const __m256i mul1 = _mm256_set1_epi16(0x0401), mul2 = _mm256_set1_epi32(0x00100001);
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                            0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    for (uint32_t i = 0; i < full; i += 32) {
        const __m256i x = _mm256_loadu_si256((const __m256i*)(src + i));
        const __m256i c = _mm256_and_si256(_mm256_srli_epi16(x, 1), three);
        ok = _mm256_and_si256(ok, _mm256_cmpeq_epi8(_mm256_shuffle_epi8(table, c), x));
        const __m256i n = _mm256_madd_epi16(_mm256_maddubs_epi16(c, mul1), mul2);
        const __m256i g = _mm256_shuffle_epi8(n, gather);
        dst[(i >> 4)] = (uint32_t)_mm256_cvtsi256_si32(g);
        dst[(i >> 4) + 1] = (uint32_t)_mm256_extract_epi32(g, 4);
    }
    const uint32_t r = len - full;
    if (r != 0 && avail >= (size_t)full + 32) {
        /* the bytes past the end count as 'A' */
        static const unsigned char keep_tab[64] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF,
                                                   0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
        const __m256i keep = _mm256_loadu_si256((const __m256i*)(keep_tab + 32 - r));
        const __m256i x = _mm256_blendv_epi8(_mm256_set1_epi8('A'), _mm256_loadu_si256((const __m256i*)(src + full)), keep);
        const __m256i c = _mm256_and_si256(_mm256_srli_epi16(x, 1), three);
        ok = _mm256_and_si256(ok, _mm256_cmpeq_epi8(_mm256_shuffle_epi8(table, c), x));
        const __m256i g = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(c, mul1), mul2), gather);
        uint32_t* out = dst + (full >> 4);
        out[0] = (uint32_t)_mm256_cvtsi256_si32(g);
        if (r > 16) { out[1] = (uint32_t)_mm256_extract_epi32(g, 4); out[2] = 0; }
        else out[1] = 0;
        return _mm256_movemask_epi8(ok) != -1;
    }
    int bad = _mm256_movemask_epi8(ok) != -1;
    return bad | pack_scalar(src, full, len, dst);
}
#endif

static int have_avx2(void) {
#if defined(__x86_64__)
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("avx2") ? 1 : 0;
    return have;
#else
    return 0;
#endif
}

int wfagpu_host_pack_sequence(const char* src, uint32_t len, uint32_t* dst) {
#if defined(__x86_64__)
    if (have_avx2()) return pack_avx2((const unsigned char*)src, len, len, dst);
#endif
    return pack_scalar((const unsigned char*)src, 0, len, dst);
}

int wfagpu_host_pack_sequence_scalar(const char* src, uint32_t len, uint32_t* dst) {
    return pack_scalar((const unsigned char*)src, 0, len, dst);
}

/* A strip of records: assigns the packed offsets from first_off on (wfagpu_amd_fill_packed_offsets' assignment) and, with a
 * staging buffer, packs the sequences there (until the first byte outside ACGT: the batch then goes up as ASCII). */
int wfagpu_host_pack_strip(const char* seq, size_t seq_bytes, sequence_pair_t* meta, size_t n, size_t first_off, uint32_t* stage) {
    int bad = 0;
    size_t off = first_off;
    const int vec = have_avx2();
    for (size_t j = 0; j < n; ++j) {
        sequence_pair_t* m = &meta[j];
        const size_t po = m->pattern_offset, to = m->text_offset;
        const uint32_t pl = (uint32_t)m->pattern_len, tl = (uint32_t)m->text_len;
        m->pattern_offset_packed = off;
        /* (a record that points outside the caller's buffer is not touched here: the batch goes up as it is) */
        if (po > seq_bytes || pl > seq_bytes - po || to > seq_bytes || tl > seq_bytes - to) bad = 1;
        if (stage && !bad) {
#if defined(__x86_64__)
            if (vec) bad |= pack_avx2((const unsigned char*)seq + po, pl, po <= seq_bytes ? seq_bytes - po : pl, stage + (off >> 2));
            else
#endif
                bad |= pack_scalar((const unsigned char*)seq + po, 0, pl, stage + (off >> 2));
        }
        off += 4 * (((size_t)pl + 15) / 16 + 1);
        m->text_offset_packed = off;
        if (stage && !bad) {
#if defined(__x86_64__)
            if (vec) bad |= pack_avx2((const unsigned char*)seq + to, tl, to <= seq_bytes ? seq_bytes - to : tl, stage + (off >> 2));
            else
#endif
                bad |= pack_scalar((const unsigned char*)seq + to, 0, tl, stage + (off >> 2));
        }
        off += 4 * (((size_t)tl + 15) / 16 + 1);
    }
    (void)vec;
    return bad;
}
