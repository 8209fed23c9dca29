/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Thin driver over the reference's own CPU ground truth (WFA2-lib v2.3,
 * vendored at /root/reference/external/WFA).  oracle/Makefile compiles the
 * reference's .c files *where they lie* together with this file into
 * oracle/_ref/libwfa2ref.so; no reference source is copied into this
 * repository.  It configures the aligner exactly as the reference's shim
 * does (utils/wfa_cpu.c:40-48 and :172-183: gap_affine, match 0, heuristic
 * none, memory mode low for the batch fallback / default for the checker).
 *
 * Used by tests/ to pin oracle/wfa_oracle.c, by tests/golden/make_golden.py
 * to produce fixtures, and by bench.py as cpu_baseline kind "reference".
 */
#include <stdint.h>
#include <stdio.h>
#include <stdbool.h>
#include <stdlib.h>
#include <string.h>

#include "wavefront/wavefront_align.h"
#include "alignment/cigar.h"

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  wavefront_aligner_t* wf;
} ref_handle_t;

/* memory_mode: 0 = high (full wavefronts, offset backtrace),
 *              1 = low  (piggy-backed backtrace; what utils/wfa_cpu.c:46 sets) */
void* ref_new(int x, int o, int e, int memory_mode, int score_only) {
  wavefront_aligner_attr_t attr = wavefront_aligner_attr_default;
  attr.distance_metric = gap_affine;
  attr.affine_penalties.match = 0;
  attr.affine_penalties.mismatch = x;
  attr.affine_penalties.gap_opening = o;
  attr.affine_penalties.gap_extension = e;
  attr.heuristic.strategy = wf_heuristic_none;
  attr.memory_mode = memory_mode ? wavefront_memory_low : wavefront_memory_high;
  attr.alignment_scope = score_only ? compute_score : compute_alignment;
  ref_handle_t* h = (ref_handle_t*)malloc(sizeof(*h));
  h->wf = wavefront_aligner_new(&attr);
  return h;
}

void ref_delete(void* hv) {
  ref_handle_t* h = (ref_handle_t*)hv;
  if (!h) return;
  wavefront_aligner_delete(h->wf);
  free(h);
}

/* Returns the positive score (utils/wfa_cpu.c:69-72 negates WFA2's). */
int ref_run(void* hv, const char* pattern, int plen, const char* text, int tlen,
            char* cigar_out, size_t cigar_cap) {
  ref_handle_t* h = (ref_handle_t*)hv;
  wavefront_align(h->wf, pattern, plen, text, tlen);
  const int score = -h->wf->cigar->score;
  if (cigar_out) {
    const size_t len = (size_t)(h->wf->cigar->end_offset - h->wf->cigar->begin_offset);
    /* worst case of "%d%c" per op run is bounded by 2 chars per op for
     * run length 1; longer runs only shrink it */
    if (2 * len + 2 > cigar_cap) return -3;
    cigar_sprint(cigar_out, h->wf->cigar, true);
  }
  return score;
}

/* Batch driver over the WFA-GPU buffer layout (see oracle_batch). */
int64_t ref_batch(const char* seqbuf, const int64_t* offsets, int64_t n,
                  int x, int o, int e, int memory_mode, int32_t* scores,
                  char* cigar_buf, size_t cigar_stride, int nthreads) {
  int64_t done = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : done)
#endif
  {
    void* h = ref_new(x, o, e, memory_mode, cigar_buf == NULL);
    /* (pairs in chunks of 16 -- of ONE when there are few of them: 46 pairs of 30 kbp in chunks of 16 kept 3 of 16 threads busy) */
    const int chunk = n >= 4096 ? 16 : 1;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, chunk)
#endif
    for (int64_t i = 0; i < n; ++i) {
      const char* p = seqbuf + offsets[4 * i + 0];
      const int plen = (int)offsets[4 * i + 1];
      const char* t = seqbuf + offsets[4 * i + 2];
      const int tlen = (int)offsets[4 * i + 3];
      const int sc = ref_run(h, p, plen, t, tlen,
                             cigar_buf ? cigar_buf + (size_t)i * cigar_stride : NULL,
                             cigar_stride);
      if (scores) scores[i] = sc;
      ++done;
    }
    ref_delete(h);
  }
  (void)nthreads;
  return done;
}
