// The C API from C++ (reference: examples/auto_example.cpp).
#include <cstdio>
#include <string>
#include <vector>
#include "include/wfa_gpu.h"

int main() {
    wfagpu_aligner_t aligner = {};
    if (!wfagpu_initialize_aligner(&aligner)) return 1;
    const std::vector<std::pair<std::string, std::string>> pairs = {
        {"ACGGTCATTCAGGATCCA", "ACGTCATTCAGGTATCCA"}, {"TTTTTTTTAAAAAAAACCCCCCCC", "TTTTTTTTCCCCCCCC"}};
    for (const auto& p : pairs) wfagpu_add_sequences(&aligner, p.first.c_str(), p.second.c_str());
    affine_penalties_t penalties = {2, 3, 1};
    if (!wfagpu_initialize_parameters(&aligner, penalties)) return 1;
    aligner.alignment_options.compute_cigar = true;
    if (!wfagpu_align(&aligner)) return 1;
    for (size_t i = 0; i < aligner.num_sequence_pairs; i++)
        std::printf("pair %zu: score %u cigar %s\n", i, aligner.results[i].error, aligner.results[i].cigar.buffer);
    wfagpu_destroy_aligner(&aligner);
    return 0;
}
