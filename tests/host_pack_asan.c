/* AddressSanitizer harness for utils/host_pack.c (built and run by tests/test_oracle.py): random records laid out back to
 * back in heap buffers of EXACTLY the bytes they need -- a read past the last base or a write past a sequence's zero word lands
 * in a red zone.  Every word is compared with a byte-at-a-time packer. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/wfa_gpu_device.h"

static uint32_t rnd_state = 12345;
static uint32_t rnd(void) { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

int main(void) {
    const char* alphabet = "ACGT";
    for (int round = 0; round < 400; ++round) {
        const int n = 1 + (int)(rnd() % 12);
        sequence_pair_t* meta = calloc((size_t)n, sizeof(*meta));
        size_t bytes = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned pl = rnd() % (round % 3 == 0 ? 40 : 400), tl = rnd() % (round % 5 == 0 ? 33 : 300);
            meta[i].pattern_len = pl; meta[i].text_len = tl;
            meta[i].pattern_offset = bytes; bytes += pl + (rnd() % 3);      /* 0-2 bytes of padding, as callers differ */
            meta[i].text_offset = bytes; bytes += tl + (i + 1 < n ? rnd() % 3 : 0);
        }
        char* seq = malloc(bytes ? bytes : 1);
        for (size_t b = 0; b < bytes; ++b) seq[b] = alphabet[rnd() & 3];
        const int dirty = round % 7 == 0 && bytes;
        size_t dirty_at = 0;
        if (dirty) {
            const int i = (int)(rnd() % n);
            if (meta[i].pattern_len) { dirty_at = meta[i].pattern_offset + rnd() % meta[i].pattern_len; seq[dirty_at] = 'N'; }
        }
        sequence_pair_t* want = malloc((size_t)n * sizeof(*want));
        memcpy(want, meta, (size_t)n * sizeof(*want));
        size_t total = 0;       /* (wfagpu_amd_fill_packed_offsets' assignment, csrc/wfa_host.hip, restated: that file needs hipcc) */
        for (int i = 0; i < n; ++i) {
            want[i].pattern_offset_packed = total; total += 4 * (((size_t)want[i].pattern_len + 15) / 16 + 1);
            want[i].text_offset_packed = total; total += 4 * (((size_t)want[i].text_len + 15) / 16 + 1);
        }
        uint32_t* words = malloc(total ? total : 4);      /* exactly the packed bytes */
        memset(words, 0xAB, total);
        const int bad = wfagpu_host_pack_strip(seq, bytes, meta, (size_t)n, 0, words);
        int really_bad = 0;
        for (size_t b = 0; b < bytes; ++b) if (!strchr("ACGT", seq[b])) {
            /* (padding bytes are letters here, so any non-letter is the planted one -- inside a sequence) */
            really_bad = 1;
        }
        if (bad != really_bad) { fprintf(stderr, "round %d: flag %d, expected %d\n", round, bad, really_bad); return 1; }
        for (int i = 0; i < n; ++i) {
            if (meta[i].pattern_offset_packed != want[i].pattern_offset_packed || meta[i].text_offset_packed != want[i].text_offset_packed) {
                fprintf(stderr, "round %d: offsets of record %d differ\n", round, i); return 1;
            }
            if (bad) continue;
            for (int which = 0; which < 2; ++which) {
                const char* s = seq + (which ? meta[i].text_offset : meta[i].pattern_offset);
                const unsigned len = which ? meta[i].text_len : meta[i].pattern_len;
                const uint32_t* w = words + (which ? meta[i].text_offset_packed : meta[i].pattern_offset_packed) / 4;
                const unsigned nw = (len + 15) / 16;
                for (unsigned k = 0; k <= nw; ++k) {
                    uint32_t word = 0;
                    for (unsigned j = 0; j < 16 && 16 * k + j < len; ++j) word |= (uint32_t)((s[16 * k + j] >> 1) & 3) << (2 * j);
                    if (w[k] != word) { fprintf(stderr, "round %d record %d seq %d word %u: %08x != %08x\n", round, i, which, k, w[k], word); return 1; }
                }
            }
        }
        /* single sequences through the public entry points (no bytes readable beyond len) */
        for (int i = 0; i < n && !bad; ++i) {
            const unsigned len = meta[i].text_len;
            char* one = malloc(len ? len : 1);
            memcpy(one, seq + meta[i].text_offset, len);
            uint32_t* out = malloc(4 * ((len + 15) / 16 + 1));
            if (wfagpu_host_pack_sequence(one, len, out) || wfagpu_host_pack_sequence_scalar(one, len, out)) { fprintf(stderr, "round %d: clean sequence flagged\n", round); return 1; }
            free(out); free(one);
        }
        free(words); free(want); free(seq); free(meta);
    }
    puts("host_pack_asan ok");
    return 0;
}
