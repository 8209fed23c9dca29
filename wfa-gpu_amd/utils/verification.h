/* Checkers behind the CLI's -c flag.  They restate what the reference's -c
 * path verifies (lib/align.cu:258-326, utils/verification.c:27-146): the
 * CIGAR replays onto both sequences, its gap-affine cost equals the reported
 * score, and the score equals an independent CPU computation.  Checker
 * only: nothing here ever produces a result the library returns. */
#ifndef WFAGPU_VERIFICATION_H
#define WFAGPU_VERIFICATION_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RLE CIGAR ("12M1X3I...") replays exactly onto pattern and text. */
bool check_cigar_edit(const char* text, const char* pattern, size_t tlen, size_t plen,
                      const char* cigar);

/* Sum of penalties of the CIGAR equals `distance`. */
bool check_affine_distance(const char* text, const char* pattern, size_t tlen, size_t plen,
                           int distance, int x, int o, int e, const char* cigar);

/* Optimal gap-affine score by a scalar dynamic program over (score,
 * diagonal); O(score * width) time.  Independent of the GPU kernels.
 * The working memory belongs to the caller: one zero-initialised
 * verification_scratch_t per worker thread, reused across pairs (it only
 * grows) and released with verification_scratch_free -- nothing is kept in
 * thread-local storage, so short-lived worker threads leak nothing. */
typedef struct { int* buf; size_t cap; } verification_scratch_t;
int verification_cpu_score_scratch(const char* pattern, const char* text, size_t plen, size_t tlen,
                                   int x, int o, int e, verification_scratch_t* scratch);
void verification_scratch_free(verification_scratch_t* scratch);
/* One-off form: allocates and frees its own scratch. */
int verification_cpu_score(const char* pattern, const char* text, size_t plen, size_t tlen,
                           int x, int o, int e);

#ifdef __cplusplus
}
#endif
#endif
