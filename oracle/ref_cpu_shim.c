/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Driver over the reference's own CPU path AS THE REFERENCE CALLS IT: utils/wfa_cpu.c
 * (compute_alignments_cpu_threaded :30-112, compute_distance_cpu_threaded :115-164 -- one WFA2 aligner per OpenMP
 * thread, memory mode low, schedule(static) over the pairs of a batch) together with utils/cigar.c, compiled by
 * oracle/Makefile from the sources where they lie under /root/reference into oracle/_ref/libwfacpuref.so, next to
 * the WFA2 sources they call.  No reference source is copied into this repository; the reference's headers are
 * included from their own tree (-I$(REF) -I$(REF)/lib).
 *
 * This is the function BASELINE.json's north_star names as the CPU baseline ("next to utils/wfa_cpu.c timed on the
 * GPU box's own host cores"): bench.py times it as cpu_baseline.reference_shim (kind "reference-shim"); the tests
 * check that it returns what oracle/_ref/libwfa2ref.so (ref_shim.c) returns.
 *
 * Every pair is handed over as "not finished by the GPU" (results[i].finished = false), which is the branch of
 * compute_alignments_cpu_threaded that aligns on the CPU (:59-84).  The caller's CIGAR buffers are sized so that the
 * function's realloc branch (:73-79, which grows a buffer to the number of operations + 1 although the RLE text of a
 * CIGAR without long runs needs two characters per operation) is never taken.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "utils/wfa_cpu.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* meta: n records in the reference layout (utils/sequences.h:28-36, 48 bytes each).  scores: int32[n], positive.
 * cigar_buf (NULL: distance only): n slots of cigar_stride bytes, NUL-terminated RLE text each.  Returns the number of
 * alignments the reference's function reports as computed on the CPU (n), or -1 when memory ran out. */
int64_t refcpu_batch(char* seqbuf, const sequence_pair_t* meta, int64_t n, int x, int o, int e, int32_t* scores,
                     char* cigar_buf, size_t cigar_stride, int nthreads) {
  if (n <= 0) return 0;
  alignment_result_t* results = (alignment_result_t*)calloc((size_t)n, sizeof(alignment_result_t));
  wfa_alignment_result_t* out = (wfa_alignment_result_t*)calloc((size_t)n, sizeof(wfa_alignment_result_t));
  if (!results || !out) { free(results); free(out); return -1; }
  for (int64_t i = 0; i < n; ++i) {
    results[i].finished = false;
    if (cigar_buf) {
      out[i].cigar.buffer = cigar_buf + (size_t)i * cigar_stride;
      out[i].cigar.buffer_size = cigar_stride;
      out[i].cigar.buffer[0] = '\0';
    }
  }
#ifdef _OPENMP
  omp_set_num_threads(nthreads > 0 ? nthreads : 1);
#endif
  int done;
  if (cigar_buf) done = compute_alignments_cpu_threaded((int)n, 0, results, out, meta, seqbuf, NULL, 0u, x, o, e, false);
  else done = compute_distance_cpu_threaded((int)n, 0, results, out, meta, seqbuf, x, o, e, false);
  if (scores) for (int64_t i = 0; i < n; ++i) scores[i] = (int32_t)out[i].error;
  free(results); free(out);
  (void)nthreads;
  return done;
}
