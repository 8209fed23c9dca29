/* Result array allocation (reference: lib/alignment_results.c:24-55). */
#include "../../include/wfa_gpu_abi.h"

bool initialize_wfa_results(wfa_alignment_result_t** results, const size_t num_alignments,
                            const size_t cigar_length) {
    if (results == NULL) return false;
    wfa_alignment_result_t* r = (wfa_alignment_result_t*)calloc(num_alignments ? num_alignments : 1, sizeof(*r));
    if (r == NULL) return false;
    *results = r;
    const size_t bytes = cigar_length ? cigar_length : 1;
    for (size_t i = 0; i < num_alignments; ++i) {
        r[i].cigar.buffer = (char*)calloc(bytes, 1);
        if (r[i].cigar.buffer == NULL) return false;
        r[i].cigar.buffer_size = bytes;
    }
    return true;
}

bool destroy_wfa_results(wfa_alignment_result_t* results, const size_t num_alignments) {
    if (results == NULL) return false;
    for (size_t i = 0; i < num_alignments; ++i) free(results[i].cigar.buffer);
    free(results);
    return true;
}
