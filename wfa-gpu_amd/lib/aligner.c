/*
 * Public C API of the aligner object -- the MI355X build's counterpart of
 * the reference's lib/aligner.c:114-263.  Same six entry points, same
 * struct (include/wfa_gpu_abi.h), same validation and return values.
 *
 * Storage: one growable byte buffer holding every pattern/text (each
 * starting on a 4-byte boundary and followed by at least one NUL) plus one
 * growable array of 48-byte records.  Unlike the reference, memory added
 * by a grow step is zeroed and the record capacity is tracked in elements
 * (SURVEY.md Appendix C lists the reference's defects there).
 */
#include <string.h>

#include "../../include/wfa_gpu_abi.h"
#include "../utils/logger.h"

#define SEQ_BUF_STEP ((size_t)1 << 20)   /* lib/aligner.c:27-29: 1 MiB, 10000 records, 50-byte CIGARs */
#define SEQ_META_STEP ((size_t)10000)
#define CIGAR_INITIAL_BYTES ((size_t)50)

static bool reserve_sequence_bytes(wfagpu_aligner_t* al, size_t needed) {
    if (needed < al->sequences_buffer_len) return true;
    size_t new_len = al->sequences_buffer_len;
    while (needed >= new_len) new_len += SEQ_BUF_STEP;
    wfagpu_seqbuf_t* nb = (wfagpu_seqbuf_t*)realloc(al->sequences_buffer, new_len);
    if (nb == NULL) {
        LOG_ERROR("Can now grow sequences buffer (realloc failed).")
        return false;
    }
    memset(nb + al->sequences_buffer_len, 0, new_len - al->sequences_buffer_len);
    al->sequences_buffer = nb;
    al->sequences_buffer_len = new_len;
    return true;
}

static bool reserve_metadata(wfagpu_aligner_t* al, size_t needed) {
    if (needed <= al->sequences_metadata_len) return true;
    size_t new_len = al->sequences_metadata_len;
    while (needed > new_len) new_len += SEQ_META_STEP;
    sequence_pair_t* nb = (sequence_pair_t*)realloc(al->sequences_metadata, new_len * sizeof(sequence_pair_t));
    if (nb == NULL) {
        LOG_ERROR("Can now grow sequences metadata buffer (realloc failed).")
        return false;
    }
    memset(nb + al->sequences_metadata_len, 0, (new_len - al->sequences_metadata_len) * sizeof(sequence_pair_t));
    al->sequences_metadata = nb;
    al->sequences_metadata_len = new_len;
    return true;
}

bool wfagpu_initialize_aligner(wfagpu_aligner_t* aligner) {
    if (aligner == NULL) {
        LOG_ERROR("Invalid aligner.")
        return false;
    }
    memset(aligner, 0, sizeof(*aligner));
    aligner->last_sequence_pair_idx = -1;
    aligner->sequences_buffer = (wfagpu_seqbuf_t*)calloc(SEQ_BUF_STEP, 1);
    aligner->sequences_metadata = (sequence_pair_t*)calloc(SEQ_META_STEP, sizeof(sequence_pair_t));
    if (aligner->sequences_buffer == NULL || aligner->sequences_metadata == NULL) {
        LOG_ERROR("Can not initialize sequences buffer.")
        free(aligner->sequences_buffer);
        free(aligner->sequences_metadata);
        aligner->sequences_buffer = NULL;
        aligner->sequences_metadata = NULL;
        return false;
    }
    aligner->sequences_buffer_len = SEQ_BUF_STEP;
    aligner->sequences_metadata_len = SEQ_META_STEP;
    return true;
}

bool wfagpu_add_sequences(wfagpu_aligner_t* aligner, const char* query, const char* target) {
    if (aligner == NULL) {
        LOG_ERROR("Invalid aligner.")
        return false;
    }
    if (query == NULL || target == NULL) {
        LOG_ERROR("Invalid sequence pointers.")
        return false;
    }
    const size_t plen = strnlen(query, MAX_SEQ_LEN);
    const size_t tlen = strnlen(target, MAX_SEQ_LEN);
    if (plen >= MAX_SEQ_LEN || tlen >= MAX_SEQ_LEN) {
        LOG_WARN("Sequences must be shorter than %lu.", MAX_SEQ_LEN - 1)
        return false;
    }
    size_t poff = 0;
    if (aligner->last_sequence_pair_idx >= 0) {
        const sequence_pair_t* last = &aligner->sequences_metadata[aligner->last_sequence_pair_idx];
        poff = WFA_ALIGN_32_BITS(last->text_offset + last->text_len + 1);
    }
    const size_t toff = WFA_ALIGN_32_BITS(poff + plen + 1);
    /* keep a full zero word after the text so that 4-byte readers never run off the end */
    if (!reserve_sequence_bytes(aligner, toff + tlen + 8)) {
        LOG_ERROR("Sequences do not fit in memory. Aborting.")
        return false;
    }
    const size_t idx = (size_t)(aligner->last_sequence_pair_idx + 1);
    if (!reserve_metadata(aligner, idx + 1)) {
        LOG_ERROR("Can not resize sequence metadata buffer. Aborting.")
        return false;
    }
    memcpy(aligner->sequences_buffer + poff, query, plen);
    memcpy(aligner->sequences_buffer + toff, target, tlen);
    sequence_pair_t* m = &aligner->sequences_metadata[idx];
    memset(m, 0, sizeof(*m));
    m->pattern_offset = poff;
    m->pattern_len = (unsigned int)plen;
    m->text_offset = toff;
    m->text_len = (unsigned int)tlen;
    aligner->last_sequence_pair_idx++;
    aligner->num_sequence_pairs++;
    return true;
}

bool wfagpu_initialize_parameters(wfagpu_aligner_t* aligner, affine_penalties_t penalties) {
    if (aligner == NULL) {
        LOG_ERROR("Invalid aligner.")
        return false;
    }
    if (penalties.x < 0 || penalties.o < 0 || penalties.e < 0) {
        LOG_ERROR("Penalties must be >= 0.")
        return false;
    }
    if (penalties.x == 0 && penalties.o == 0 && penalties.e == 0) {
        LOG_ERROR("All penalties can not be 0.")
        return false;
    }
    if (penalties.x == 0 || penalties.e == 0) {
        /* WFA needs strictly positive mismatch and extension costs (WFA2 rejects
         * them too: external/WFA/wavefront/wavefront_penalties.c:96-105). */
        LOG_ERROR("Mismatch and gap-extension penalties must be > 0.")
        return false;
    }
    if (aligner->num_sequence_pairs == 0) {
        LOG_ERROR("No sequences added.")
        return false;
    }
    wfagpu_set_default_options(&aligner->alignment_options, aligner->sequences_metadata, penalties,
                               aligner->num_sequence_pairs);
    if (aligner->results != NULL) destroy_wfa_results(aligner->results, aligner->num_sequence_pairs);
    aligner->results = NULL;
    return initialize_wfa_results(&aligner->results, aligner->num_sequence_pairs, CIGAR_INITIAL_BYTES);
}

bool wfagpu_set_batch_size(wfagpu_aligner_t* aligner, size_t batch_size) {
    if (aligner == NULL) {
        LOG_ERROR("Invalid aligner.")
        return false;
    }
    if (batch_size > aligner->num_sequence_pairs) {
        LOG_WARN("Batch size must be less or equal than the number of sequences. Setting batch size to %lu.",
                 aligner->num_sequence_pairs)
        batch_size = aligner->num_sequence_pairs;
    }
    if (batch_size == 0) {
        LOG_WARN("Batch size can not be zero. Setting batch size to %lu.", aligner->num_sequence_pairs)
        batch_size = aligner->num_sequence_pairs;
    }
    aligner->alignment_options.batch_size = batch_size;
    return true;
}

size_t wfagpu_amd_results_capacity(const wfa_alignment_result_t* results);      /* lib/alignment_results.c */

bool wfagpu_align(wfagpu_aligner_t* aligner) {
    if (aligner == NULL) {
        LOG_ERROR("Invalid aligner.")
        return false;
    }
    /* (an aligner without sequences: the reference calls the launcher all the same and returns true,
     * lib/aligner.c:236-263 -- this build's launcher does nothing for zero alignments, same return value.  Sequences but no
     * results array -- wfagpu_initialize_parameters was never called, so the options are not initialised either: the
     * reference dereferences the missing array; here that is an error the caller gets told about) */
    if (aligner->results == NULL && aligner->num_sequence_pairs > 0) {
        LOG_ERROR("No results array: call wfagpu_initialize_parameters before wfagpu_align.")
        return false;
    }
    /* sequences added after the parameters were initialised: the results array is still the old count's (the reference
     * indexes it by the new one) -- a fresh array for every pair */
    if (aligner->results != NULL && aligner->num_sequence_pairs > 0) {
        const size_t have = wfagpu_amd_results_capacity(aligner->results);
        if (have != (size_t)-1 && have < aligner->num_sequence_pairs) {
            destroy_wfa_results(aligner->results, have);
            aligner->results = NULL;
            if (!initialize_wfa_results(&aligner->results, aligner->num_sequence_pairs, CIGAR_INITIAL_BYTES)) {
                LOG_ERROR("Can not allocate the results.")
                return false;
            }
        }
    }
    /* callers set options by poking the struct (examples/manual_example.c:67-91) */
    aligner->alignment_options.num_alignments = aligner->num_sequence_pairs;
    if (aligner->alignment_options.compute_cigar) {
        launch_alignments(aligner->sequences_buffer, aligner->sequences_buffer_len, aligner->sequences_metadata,
                          aligner->results, aligner->alignment_options, false);
    } else {
        launch_alignments_distance(aligner->sequences_buffer, aligner->sequences_buffer_len,
                                   aligner->sequences_metadata, aligner->results, aligner->alignment_options, false);
    }
    return true;
}

void wfagpu_destroy_aligner(wfagpu_aligner_t* aligner) {
    if (aligner == NULL) return;
    free(aligner->sequences_buffer);
    free(aligner->sequences_metadata);
    if (aligner->results != NULL) destroy_wfa_results(aligner->results, aligner->num_sequence_pairs);
    aligner->sequences_buffer = NULL;
    aligner->sequences_metadata = NULL;
    aligner->results = NULL;
}
