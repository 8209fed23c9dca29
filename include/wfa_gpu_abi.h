/*
 * wfa_gpu_abi.h -- the drop-in boundary of the MI355X build.
 *
 * Everything a program written against quim0/WFA-GPU's library compiles and
 * links against, with the same names, argument meaning, struct layouts and
 * error behaviour.  Each item cites the reference interface it replaces
 * (paths relative to the reference tree).  The thin headers under
 * wfa-gpu_amd/lib and wfa-gpu_amd/utils carry the reference's header names
 * (lib/include/wfa_gpu.h, lib/aligner.h, ...) and simply include this file,
 * so `#include "include/wfa_gpu.h"` with `-I lib -I .` keeps working
 * (README.md:106, examples/Makefile:8 of the reference).
 *
 * Plain C, no HIP or torch types anywhere in the signatures.
 */
#ifndef WFA_GPU_ABI_H
#define WFA_GPU_ABI_H

#include <inttypes.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types */

/* utils/sequences.h:28-36 -- one record per pair, 48 bytes.  Offsets are
 * byte offsets into the sequence buffer; every sequence starts 4-byte
 * aligned and is followed by 1..4 NUL bytes. */
typedef struct {
    size_t text_offset;
    size_t pattern_offset;
    size_t text_offset_packed;     /* filled by launch_alignments* (lib/align.cu:103-115) */
    size_t pattern_offset_packed;
    unsigned int text_len;
    unsigned int pattern_len;
    bool has_N;                    /* input: ignored.  The reference's packing kernel sets it in its DEVICE copy
                                    * only (sequence_packing_kernel.cu:57,74; no D2H of the records); this build
                                    * keeps the same information in a device-side flag array, so the caller's
                                    * copy is left as it was in both implementations. */
} sequence_pair_t;

/* lib/affine_penalties.h:25-30 -- match cost is 0 by construction */
typedef struct {
    int x;   /* mismatch      (> 0)  */
    int o;   /* gap open      (>= 0) */
    int e;   /* gap extension (> 0)  */
} affine_penalties_t;

/* lib/wfa_types.h:31-64.  Kept because callers see them; this build's
 * kernels do not use the piggy-back block format. */
#define MAX_SEQ_LEN (1UL << 15)
typedef int16_t wfa_offset_t;
#define wfa_backtrace_bits 32
typedef uint32_t bt_vector_t;
typedef uint32_t bt_prev_t;
typedef struct {
    bt_vector_t backtrace;
    bt_prev_t prev;
} wfa_backtrace_t;
typedef enum { OP_NOOP = 0, OP_INS = 1, OP_SUB = 2, OP_DEL = 3 } affine_op_t;
static const char ops_ascii[4] = {'?', 'I', 'X', 'D'};
typedef enum { GAP_OPEN = 1, GAP_EXTEND } gap_op_t;
#define BT_OFFLOADED_ELEMENTS(max_steps) \
    (((max_steps) * 2 + 1) * ((max_steps) * 2 / (wfa_backtrace_bits / 2)))
#define BT_OFFLOADED_RESULT_ELEMENTS(max_steps) ((max_steps) * 2 / (wfa_backtrace_bits / 2))

/* lib/alignment_results.h:30-48 */
typedef struct {
    char* buffer;              /* NUL-terminated RLE CIGAR: nM nX nI nD        */
    size_t buffer_size;
    size_t last_free_position;
} wfa_cigar_t;

typedef struct {               /* internal record of the reference's kernels   */
    bool finished;
    int distance;
    wfa_backtrace_t backtrace;
    int num_bt_blocks;
} alignment_result_t;

typedef struct {
    unsigned int error;        /* positive gap-affine score                    */
    wfa_cigar_t cigar;
} wfa_alignment_result_t;

/* lib/alignment_parameters.h:29-58 */
#define BAND_NONE (-1)
#ifndef MAX
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#endif
#ifndef MIN
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#endif

typedef struct {
    int max_error;             /* expected score ceiling; sizes the first kernel tier.
                                  Pairs that exceed it are re-run on the GPU in a
                                  wider tier (the reference re-ran them on the CPU). */
    int threads_per_block;     /* accepted for compatibility; also the band width in banded mode */
    int num_workers;           /* accepted for compatibility (the build sizes its own grids)     */
    int band;                  /* BAND_NONE or the re-centring period of the adaptive band       */
    size_t batch_size;
    size_t num_alignments;
    affine_penalties_t penalties;
    bool compute_cigar;
} wfa_alignment_options_t;

/* lib/aligner.h:30-47 */
#define WFA_ALIGN_32_BITS(x) ((x) + (4 - ((x) % 4)))
typedef char wfagpu_seqbuf_t;

typedef struct {
    wfagpu_seqbuf_t* sequences_buffer;
    size_t sequences_buffer_len;
    sequence_pair_t* sequences_metadata;
    size_t sequences_metadata_len;
    size_t num_sequence_pairs;
    wfa_alignment_result_t* results;
    int64_t last_sequence_pair_idx;
    wfa_alignment_options_t alignment_options;
} wfagpu_aligner_t;

/* ---------------------------------------------------- device query shims */

/* utils/device_query.cuh:29-33.  Names kept (the CLI and the static helper
 * below link against them); implemented on hipGetDeviceProperties. */
void get_num_cuda_devices(int* n);
char* get_cuda_dev_name(int dev);          /* caller frees */
int get_cuda_SM_count(int dev);            /* compute units on AMD */
void get_cuda_capability(int dev, int* major, int* minor);

/* ------------------------------------------------------ results storage */

/* lib/alignment_results.h:54-60.  An array made by initialize_wfa_results is released through destroy_wfa_results (the
 * library remembers how many records it made, so that an aligner whose pair count changed afterwards frees and indexes the
 * array by the right count); arrays built by the caller are the caller's to free. */
bool initialize_wfa_results(wfa_alignment_result_t** results,
                            const size_t num_alignments,
                            const size_t cigar_length);
bool destroy_wfa_results(wfa_alignment_result_t* results,
                         const size_t num_alignments);

/* --------------------------------------------------- the C-ABI seam (L3) */

/* lib/align.cuh:35-47.  Blocking; results in input order; mutates
 * sequences_metadata[*].{text,pattern}_offset_packed like the reference does
 * (lib/align.cu:103-115; the values are batch-relative offsets of THIS build's
 * packed layout) and nothing else of the caller's records.  With check_correctness the -c self-check of
 * lib/align.cu:258-326 / :688-739 runs and prints
 * "correct=%d Incorrect=%d" lines on stderr. */
void launch_alignments(char* sequences_buffer,
                       const size_t sequences_buffer_size,
                       sequence_pair_t* const sequences_metadata,
                       wfa_alignment_result_t* const alignment_results,
                       wfa_alignment_options_t options,
                       bool check_correctness);

void launch_alignments_distance(char* sequences_buffer,
                                const size_t sequences_buffer_size,
                                sequence_pair_t* const sequences_metadata,
                                wfa_alignment_result_t* const alignment_results,
                                wfa_alignment_options_t options,
                                bool check_correctness);

/* ------------------------------------------------------- public API (L4) */

/* lib/aligner.h:49-62 / lib/aligner.c:114-263 */
bool wfagpu_initialize_aligner(wfagpu_aligner_t* aligner);
bool wfagpu_add_sequences(wfagpu_aligner_t* aligner, const char* query, const char* target);
bool wfagpu_initialize_parameters(wfagpu_aligner_t* aligner, affine_penalties_t penalties);
bool wfagpu_set_batch_size(wfagpu_aligner_t* aligner, size_t batch_size);
bool wfagpu_align(wfagpu_aligner_t* aligner);
void wfagpu_destroy_aligner(wfagpu_aligner_t* aligner);

#ifdef __cplusplus
}
#endif

/* ------------------------------ header-level helpers callers compile in */

/* lib/alignment_parameters.h:60-71 */
static inline int wfa_get_threads_per_alignment(const size_t max_error) {
    const size_t wf = 2 * max_error + 1;
    if (wf <= 128) return 64;
    if (wf <= 256) return 128;
    if (wf <= 512) return 256;
    if (wf <= 1024) return 512;
    return 1024;
}

/* lib/alignment_parameters.h:73-81 -- on MI355X a "worker" is one
 * workgroup; the library re-derives its own grid, this only keeps user code
 * (examples/manual_example.c:84) compiling and meaningful: compute units x
 * workgroups of that size that fit a CU's 32 wave slots. */
static inline int get_num_workers(const int num_threads) {
    const int cus = get_cuda_SM_count(0);
    const int waves = (num_threads + 63) / 64;
    const int per_cu = 32 / (waves > 0 ? waves : 1);
    return cus * (per_cu > 0 ? per_cu : 1);
}

/* lib/alignment_parameters.h:83-106 */
static inline void wfagpu_set_default_options(wfa_alignment_options_t* wfa_options,
                                              sequence_pair_t* sequences_metadata,
                                              affine_penalties_t penalties,
                                              size_t num_alignments) {
    int slen = (int)MAX(sequences_metadata[0].pattern_len, sequences_metadata[0].text_len);
    slen = (int)(slen * 0.1);
    int max_error = slen * MAX(penalties.x, MAX(penalties.o, penalties.e));
    if (max_error < 50) max_error = 50;
    wfa_options->max_error = max_error;
    wfa_options->threads_per_block = wfa_get_threads_per_alignment((size_t)max_error);
    wfa_options->num_workers = get_num_workers(wfa_options->threads_per_block);
    wfa_options->band = BAND_NONE;
    wfa_options->num_alignments = num_alignments;
    wfa_options->batch_size = (num_alignments > 10) ? num_alignments / 10 : num_alignments;
    wfa_options->penalties = penalties;
    wfa_options->compute_cigar = false;
}

#endif /* WFA_GPU_ABI_H */
