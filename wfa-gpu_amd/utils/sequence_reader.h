/* Input readers of the wfa.affine.gpu CLI: ".seq" (alternating ">PATTERN" and
 * "<TEXT" lines) and paired FASTA (i-th record of the query file against the
 * i-th record of the target file, multi-line records joined, headers
 * ignored).  Same formats as the reference's utils/sequence_reader.c:137-392;
 * both fill the padded batch layout of utils/sequences.h directly. */
#ifndef WFAGPU_SEQUENCE_READER_H
#define WFAGPU_SEQUENCE_READER_H

#include "../../include/wfa_gpu_abi.h"

typedef struct {
    char* sequences_buffer;
    size_t sequences_buffer_size;     /* allocated bytes */
    size_t sequences_buffer_used;
    sequence_pair_t* sequences_metadata;
    size_t metadata_capacity;
    size_t num_pairs;
} sequence_set_t;

/* max_pairs == 0 reads everything.  Return false on I/O or format errors.  `set` must be empty (zeroed).  .seq files are
 * mapped and parsed by all the cores the process may use (sequence_reader.c). */
extern int sequence_reader_force_threads;      /* test hook: that many strips whatever the file's size (0: automatic) */
bool read_seq_file(sequence_set_t* set, const char* path, size_t max_pairs);
bool read_fasta_pair_files(sequence_set_t* set, const char* query_path, const char* target_path, size_t max_pairs);
void free_sequence_set(sequence_set_t* set);

#endif
