/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the gap-affine end-to-end WFA path that
 * quim0/WFA-GPU uses as its ground truth (its vendored WFA2-lib v2.3,
 * reached through utils/wfa_cpu.c:30-189).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this; the product
 * library (wfa-gpu_amd/libwfagpu.so) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement
 *   - against the reference's committed golden vectors (tests/golden/*.alg,
 *     copied data files of /root/reference/external/WFA/tests/wfa.utest.check
 *     and /root/reference/tests/data/results), and
 *   - against the reference's own WFA2 sources compiled in place
 *     (oracle/_ref/libwfa2ref.so, see oracle/Makefile) on seeded random pairs.
 */
#ifndef WFA_ORACLE_H
#define WFA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int64_t cells;        /* sum over computed score steps of (hi-lo+1), WFA2 ranges  */
  int64_t steps;        /* number of non-null score steps computed (score 0 included) */
  int32_t max_abs_k;    /* widest diagonal touched (pre-trim ranges)                */
  int32_t num_ops;      /* X/I/D operations in the alignment (CIGAR mode only)      */
} oracle_stats_t;

/* Opaque per-thread workspace, reused across pairs (the pattern of
 * utils/wfa_cpu.c:52-57: one aligner per OpenMP thread). */
typedef struct oracle_aligner oracle_aligner_t;

oracle_aligner_t* oracle_aligner_new(int x, int o, int e);
void oracle_aligner_delete(oracle_aligner_t* al);

/* Score only (ring of max(x,o+e)+1 wavefronts).  Returns the positive
 * gap-affine score, or -1 on bad arguments.  max_score<=0 means unbounded;
 * otherwise returns -2 if the score would exceed max_score. */
int oracle_score(oracle_aligner_t* al, const char* pattern, int plen,
                 const char* text, int tlen, int max_score,
                 oracle_stats_t* stats);

/* Score + CIGAR (all wavefronts kept; backward trace with WFA2's
 * tie-breaks).  cigar_out receives the RLE text ("12M1X3I...") and is
 * NUL-terminated; returns the score, -1 on bad arguments, -3 if cigar_cap
 * is too small. */
int oracle_align(oracle_aligner_t* al, const char* pattern, int plen,
                 const char* text, int tlen, char* cigar_out, size_t cigar_cap,
                 oracle_stats_t* stats);

/* Batch drivers over the WFA-GPU sequence buffer layout
 * (utils/sequences.h:28-36): offsets[4*i+0..3] = pattern_offset,
 * pattern_len, text_offset, text_len.  nthreads>1 uses OpenMP when compiled
 * with it.  cigar_buf may be NULL (score only); otherwise each pair gets
 * cigar_stride bytes.  Returns number of pairs done. */
int64_t oracle_batch(const char* seqbuf, const int64_t* offsets, int64_t n,
                     int x, int o, int e, int32_t* scores, char* cigar_buf,
                     size_t cigar_stride, int64_t* cells, int nthreads);

/* 2-bit packing restatement (lib/kernels/sequence_packing_kernel.cu:28-116):
 * code = (c & 6) >> 1.  This build's word layout is little-endian: base i of
 * a sequence sits in bits [2*(i%16) .. 2*(i%16)+1] of 32-bit word i/16.
 * Returns 1 if any byte is not one of 'A','C','G','T'. */
int oracle_pack2(const char* seq, int len, uint32_t* words_out);

/* The reference's -c checkers (utils/verification.c:27-146) restated:
 * returns 1 when the CIGAR replays exactly onto both sequences, and
 * computes its gap-affine cost. */
int oracle_check_cigar(const char* pattern, int plen, const char* text,
                       int tlen, const char* cigar, int x, int o, int e,
                       int* cost_out);

/* The reference's ADAPTIVE-BAND distance kernel restated deterministically (band_oracle.c; lib/kernels/
 * sequence_distance_kernel_aband.cu): beta = diagonals kept per wavefront (the reference's threads per block), lambda = the
 * re-centring period (-B), max_steps = the reference's max_steps (-e).  Returns the distance the kernel would report;
 * *finished_out = 0 when it gave up (steps exhausted) -- the reference then falls back to the CPU. */
int oracle_band_ref(const char* pattern, int plen, const char* text, int tlen, int x, int o, int e,
                    int beta, int lambda, int max_steps, int* finished_out, int* steps_out);
int64_t oracle_band_ref_batch(const char* seqbuf, const int64_t* offsets, int64_t n, int x, int o, int e, int beta, int lambda,
                              int max_steps, int32_t* scores, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
