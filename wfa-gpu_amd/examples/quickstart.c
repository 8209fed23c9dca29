/* Minimal use of the public C API (same calls as the reference's
 * examples/auto_example.c): add pairs, initialise parameters, align, read
 * scores and CIGARs. */
#include <stdio.h>
#include "include/wfa_gpu.h"

int main(void) {
    wfagpu_aligner_t aligner = {0};
    if (!wfagpu_initialize_aligner(&aligner)) return 1;

    /* wfagpu_add_sequences(aligner, query/pattern, target/text) */
    wfagpu_add_sequences(&aligner, "GATTACAGATTACAGATTACATTTGACCA", "GATTACAGATACAGATTACATTTGGACCA");
    wfagpu_add_sequences(&aligner, "ACGTACGTACGTACGTTTTTACGTACGT", "ACGTACGAACGTACGTACGTACGT");
    wfagpu_add_sequences(&aligner, "TTGACCATTGACCATTGACCA", "TTGACCATTGACCATTGACCA");

    affine_penalties_t penalties = {.x = 2, .o = 3, .e = 1};
    if (!wfagpu_initialize_parameters(&aligner, penalties)) return 1;
    if (!wfagpu_set_batch_size(&aligner, 2)) return 1;   /* 3 pairs -> two batches */
    aligner.alignment_options.compute_cigar = true;
    if (!wfagpu_align(&aligner)) return 1;

    for (size_t i = 0; i < aligner.alignment_options.num_alignments; i++)
        printf("pair %zu: score %u cigar %s\n", i, aligner.results[i].error, aligner.results[i].cigar.buffer);
    wfagpu_destroy_aligner(&aligner);
    return 0;
}
