/* Setting alignment options by hand, the way the reference's
 * examples/manual_example.c does: callers poke the options struct directly. */
#include <stdio.h>
#include "include/wfa_gpu.h"

int main(void) {
    wfagpu_aligner_t aligner = {0};
    if (!wfagpu_initialize_aligner(&aligner)) return 1;
    wfagpu_add_sequences(&aligner, "CCTAACCCTAACCCTAACCCTAAACCCTAAACC", "CCTAACCCTAACCCTAACCCTAACCCCTAACCC");
    wfagpu_add_sequences(&aligner, "GGTGAGGGTGAGGGTTAGGGTTAGG", "GGTGAGGGTGAGGGTTAGGGTGAGG");
    affine_penalties_t penalties = {.x = 4, .o = 6, .e = 2};
    if (!wfagpu_initialize_parameters(&aligner, penalties)) return 1;

    aligner.alignment_options.max_error = 20;            /* first-tier size; larger scores still finish on the GPU */
    aligner.alignment_options.threads_per_block = 128;   /* accepted for compatibility */
    aligner.alignment_options.num_workers = get_num_workers(aligner.alignment_options.threads_per_block);
    aligner.alignment_options.band = BAND_NONE;          /* exact */
    aligner.alignment_options.compute_cigar = false;     /* scores only */
    if (!wfagpu_align(&aligner)) return 1;
    for (size_t i = 0; i < aligner.num_sequence_pairs; i++) printf("pair %zu: score %u\n", i, aligner.results[i].error);
    wfagpu_destroy_aligner(&aligner);
    return 0;
}
