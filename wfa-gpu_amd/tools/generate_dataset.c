/*
 * Seeded synthetic read-pair generator.
 *
 * Restates the error model of WFA2's tools/generate_dataset
 * (external/WFA/tools/generate_dataset/generate_dataset.c:140-216,350-400 of
 * the reference tree): the text is `length` i.i.d. uniform bases over ACGT;
 * the pattern is a copy with ceil(length*error) edits applied one after the
 * other, each uniformly a mismatch (to a different base), a 1-base deletion
 * or a 1-base insertion at a uniform position of the current string.  The
 * reference tool seeds from time(0); here every pair i derives its own
 * stream from a splitmix64 mix of (seed, i), so any subset is reproducible and
 * generation is thread-parallel.
 *
 * Output is the WFA-GPU batch layout directly (utils/sequences.h:28-36,
 * lib/aligner.c:127-166): 4-byte aligned, NUL padded sequences in one byte
 * buffer plus 48-byte records.  Built both as a tool (-DGENERATE_DATASET_MAIN,
 * writes .seq text: ">pattern\n<text\n") and as libwfagen.so for bench.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/wfa_gpu_abi.h"

static inline uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint32_t rnd_below(uint64_t* s, uint32_t n) {   /* uniform in [0,n) */
    return (uint32_t)(((splitmix64(s) >> 32) * (uint64_t)n) >> 32);
}

static int num_errors_for(int length, double error) {
    return error >= 1.0 ? (int)error : (int)ceil((double)length * error);
}

static size_t pad4(size_t x) { return x + (4 - (x % 4)); }   /* WFA_ALIGN_32_BITS */

size_t wfagen_pair_stride(int length, double error) {
    const int ne = num_errors_for(length, error);
    return pad4((size_t)length + (size_t)ne + 1) + pad4((size_t)length + 1);
}

/* Fills seqbuf (zeroed by the caller or not: every byte of each slot is
 * written) and meta[n].  Returns bytes used, 0 if cap is too small. */
size_t wfagen_generate(char* seqbuf, size_t cap, sequence_pair_t* meta, size_t n, int length,
                       double error, uint64_t seed, int nthreads) {
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    const int ne = num_errors_for(length, error);
    const size_t pslot = pad4((size_t)length + (size_t)ne + 1);
    const size_t stride = wfagen_pair_stride(length, error);
    if (stride * n + 16 > cap) return 0;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (long long i = 0; i < (long long)n; ++i) {
        /* (the pair index goes through a mixing step of its own: seeding stream i with seed + i * GAMMA, GAMMA being
         *  splitmix64's own increment, made the text of pair i+1 the text of pair i shifted by one base in round 1) */
        uint64_t z = (seed * 0x100000001B3ull) ^ ((uint64_t)i * 0xD6E8FEB86659FD93ull + 0x1234567ull);
        uint64_t st = splitmix64(&z) ^ (uint64_t)i;
        char* p = seqbuf + stride * (size_t)i;
        char* t = p + pslot;
        for (int j = 0; j < length; ++j) t[j] = alphabet[rnd_below(&st, 4)];
        memset(t + length, 0, stride - pslot - (size_t)length);
        memcpy(p, t, (size_t)length);
        int len = length;
        for (int k = 0; k < ne; ++k) {
            const uint32_t type = rnd_below(&st, 3);
            if (type == 0) {                      /* mismatch */
                if (len == 0) continue;
                uint32_t pos; char c;
                do { pos = rnd_below(&st, (uint32_t)len); c = alphabet[rnd_below(&st, 4)]; } while (p[pos] == c);
                p[pos] = c;
            } else if (type == 1) {               /* deletion */
                if (len == 0) continue;
                const uint32_t pos = rnd_below(&st, (uint32_t)len);
                memmove(p + pos, p + pos + 1, (size_t)(len - 1) - pos);
                --len;
            } else {                              /* insertion */
                const uint32_t pos = len ? rnd_below(&st, (uint32_t)len) : 0;
                memmove(p + pos + 1, p + pos, (size_t)len - pos);
                p[pos] = alphabet[rnd_below(&st, 4)];
                ++len;
            }
        }
        memset(p + len, 0, pslot - (size_t)len);
        sequence_pair_t* m = &meta[i];
        memset(m, 0, sizeof(*m));
        m->pattern_offset = stride * (size_t)i;
        m->pattern_len = (unsigned int)len;
        m->text_offset = stride * (size_t)i + pslot;
        m->text_len = (unsigned int)length;
    }
    memset(seqbuf + stride * n, 0, 16);
    return stride * n + 16;
}

/* ---- long-read shaped error model ("hard" pairs for the adaptive-band study) -----------------------------------
 * The i.i.d. single-base model above is the easiest possible input for a band heuristic: no gap ever moves the optimal
 * path by more than one diagonal at a time.  Real long reads (and the data behind the reference's recall figures,
 * README.md:125-137) have indels of many bases and errors that come in clusters.  Here an error EVENT is a mismatch, an
 * insertion or a deletion; indel lengths are geometric with mean `indel_mean`, with probability `long_frac` replaced by
 * a long indel uniform in [long_min, long_max]; with probability `cluster` an event lands within +-32 bases of the
 * previous one instead of at a uniform position.  `error` is the fraction of bases hit by events, as before.
 * Every pair keeps its own splitmix64 stream (reproducible, thread-parallel).  Pattern lengths vary, so the layout is
 * built with a slot of length + max growth per pair. */
typedef struct {
    double error;        /* events per base                                   */
    double indel_frac;   /* share of events that are indels (rest: mismatches) */
    double indel_mean;   /* mean length of a short indel (geometric), >= 1     */
    double long_frac;    /* share of indels that are long                      */
    int long_min, long_max;
    double cluster;      /* probability that an event lands next to the previous one */
} wfagen_model_t;

static inline double rnd_unit(uint64_t* s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

size_t wfagen_model_stride(int length, const wfagen_model_t* m) {
    /* worst case growth: every event a long insertion is absurd; bound by 4x the expected inserted bases + slack */
    const double ev = (double)length * m->error;
    const double grow = ev * m->indel_frac * (m->indel_mean + m->long_frac * (double)m->long_max) * 4.0 + 1024.0 + (double)m->long_max;
    return pad4((size_t)length + (size_t)grow + 1) + pad4((size_t)length + 1);
}

size_t wfagen_generate_model(char* seqbuf, size_t cap, sequence_pair_t* meta, size_t n, int length,
                             const wfagen_model_t* m, uint64_t seed, int nthreads) {
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    const size_t stride = wfagen_model_stride(length, m);
    const size_t pslot = stride - pad4((size_t)length + 1);
    if (stride * n + 16 > cap) return 0;
    const int nev = (int)ceil((double)length * m->error);
    const double geo_p = m->indel_mean > 1.0 ? 1.0 / m->indel_mean : 1.0;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (long long i = 0; i < (long long)n; ++i) {
        /* (the pair index goes through a full mixing step of its own: seeding stream i with seed + i * GAMMA, GAMMA
         *  being splitmix64's own increment, would make stream i+1 the stream of pair i advanced by one draw) */
        uint64_t z = (seed * 0x100000001B3ull) ^ ((uint64_t)i * 0xD6E8FEB86659FD93ull + 0x7654321ull);
        uint64_t st = splitmix64(&z) ^ (uint64_t)i;
        char* p = seqbuf + stride * (size_t)i;
        char* t = p + pslot;
        for (int j = 0; j < length; ++j) t[j] = alphabet[rnd_below(&st, 4)];
        memset(t + length, 0, stride - pslot - (size_t)length);
        memcpy(p, t, (size_t)length);
        int len = length;
        long prev = -1;
        for (int k = 0; k < nev; ++k) {
            long pos;
            if (prev >= 0 && rnd_unit(&st) < m->cluster) {
                pos = prev + (long)rnd_below(&st, 65) - 32;
                if (pos < 0) pos = 0;
                if (pos > len) pos = len;
            } else {
                pos = (long)rnd_below(&st, (uint32_t)(len + 1));
            }
            prev = pos;
            if (rnd_unit(&st) >= m->indel_frac) {                     /* mismatch */
                if (len == 0) continue;
                if (pos >= len) pos = len - 1;
                char c;
                do { c = alphabet[rnd_below(&st, 4)]; } while (c == p[pos]);
                p[pos] = c;
                continue;
            }
            int L = 1;
            if (rnd_unit(&st) < m->long_frac) L = m->long_min + (int)rnd_below(&st, (uint32_t)(m->long_max - m->long_min + 1));
            else while (rnd_unit(&st) >= geo_p && L < 64) ++L;
            if (rnd_below(&st, 2) == 0) {                             /* deletion of L bases */
                if (pos >= len) continue;
                if (pos + L > len) L = len - (int)pos;
                memmove(p + pos, p + pos + L, (size_t)(len - pos - L));
                len -= L;
            } else {                                                  /* insertion of L random bases */
                if ((size_t)len + (size_t)L + 1 > pslot - 4) continue;
                memmove(p + pos + L, p + pos, (size_t)(len - pos));
                for (int j = 0; j < L; ++j) p[pos + j] = alphabet[rnd_below(&st, 4)];
                len += L;
            }
        }
        memset(p + len, 0, pslot - (size_t)len);
        sequence_pair_t* mm = &meta[i];
        memset(mm, 0, sizeof(*mm));
        mm->pattern_offset = stride * (size_t)i;
        mm->pattern_len = (unsigned int)len;
        mm->text_offset = stride * (size_t)i + pslot;
        mm->text_len = (unsigned int)length;
    }
    memset(seqbuf + stride * n, 0, 16);
    return stride * n + 16;
}

#ifdef GENERATE_DATASET_MAIN
int main(int argc, char** argv) {
    size_t n = 1000; int length = 1000; double error = 0.05; uint64_t seed = 1; const char* out = NULL; int threads = 1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-n") && i + 1 < argc) n = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "-l") && i + 1 < argc) length = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-e") && i + 1 < argc) error = atof(argv[++i]);
        else if (!strcmp(argv[i], "-s") && i + 1 < argc) seed = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
        else { fprintf(stderr, "usage: %s -n pairs -l length -e error -s seed [-t threads] [-o file.seq]\n", argv[0]); return 1; }
    }
    const size_t cap = wfagen_pair_stride(length, error) * n + 64;
    char* buf = (char*)calloc(cap, 1);
    sequence_pair_t* meta = (sequence_pair_t*)calloc(n ? n : 1, sizeof(*meta));
    if (!buf || !meta || !wfagen_generate(buf, cap, meta, n, length, error, seed, threads)) { fprintf(stderr, "generation failed\n"); return 1; }
    FILE* f = out ? fopen(out, "w") : stdout;
    if (!f) { perror(out); return 1; }
    for (size_t i = 0; i < n; ++i)
        fprintf(f, ">%.*s\n<%.*s\n", (int)meta[i].pattern_len, buf + meta[i].pattern_offset, (int)meta[i].text_len, buf + meta[i].text_offset);
    if (out) fclose(f);
    free(buf); free(meta);
    return 0;
}
#endif
