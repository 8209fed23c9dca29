/* See sequence_reader.h. */
#define _GNU_SOURCE
#include "sequence_reader.h"

#include <stdio.h>
#include <string.h>

#include "logger.h"

static bool set_reserve(sequence_set_t* s, size_t extra_bytes) {
    const size_t need = s->sequences_buffer_used + extra_bytes + 64;
    if (need > s->sequences_buffer_size) {
        size_t cap = s->sequences_buffer_size ? s->sequences_buffer_size : ((size_t)1 << 20);
        while (cap < need) cap *= 2;
        char* nb = (char*)realloc(s->sequences_buffer, cap);
        if (!nb) return false;
        memset(nb + s->sequences_buffer_size, 0, cap - s->sequences_buffer_size);
        s->sequences_buffer = nb;
        s->sequences_buffer_size = cap;
    }
    if (s->num_pairs + 1 > s->metadata_capacity) {
        size_t cap = s->metadata_capacity ? s->metadata_capacity * 2 : 4096;
        sequence_pair_t* nm = (sequence_pair_t*)realloc(s->sequences_metadata, cap * sizeof(*nm));
        if (!nm) return false;
        memset(nm + s->metadata_capacity, 0, (cap - s->metadata_capacity) * sizeof(*nm));
        s->sequences_metadata = nm;
        s->metadata_capacity = cap;
    }
    return true;
}

/* Appends one pair in the layout of lib/aligner.c:127-166. */
static bool set_append(sequence_set_t* s, const char* pattern, size_t plen, const char* text, size_t tlen) {
    if (plen >= MAX_SEQ_LEN || tlen >= MAX_SEQ_LEN) {
        LOG_ERROR("Sequence %zu is too long (max %lu bases).", s->num_pairs, MAX_SEQ_LEN - 1)
        return false;
    }
    if (!set_reserve(s, plen + tlen + 16)) return false;
    const size_t poff = s->sequences_buffer_used;          /* kept 4-byte aligned */
    const size_t toff = WFA_ALIGN_32_BITS(poff + plen + 1);
    memcpy(s->sequences_buffer + poff, pattern, plen);
    memcpy(s->sequences_buffer + toff, text, tlen);
    sequence_pair_t* m = &s->sequences_metadata[s->num_pairs++];
    memset(m, 0, sizeof(*m));
    m->pattern_offset = poff; m->pattern_len = (unsigned int)plen;
    m->text_offset = toff; m->text_len = (unsigned int)tlen;
    for (size_t i = 0; i < plen && !m->has_N; ++i) m->has_N = (pattern[i] == 'N');
    for (size_t i = 0; i < tlen && !m->has_N; ++i) m->has_N = (text[i] == 'N');
    s->sequences_buffer_used = WFA_ALIGN_32_BITS(toff + tlen + 1);
    return true;
}

static size_t chomp(char* line, ssize_t len) {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = '\0';
    return (size_t)len;
}

bool read_seq_file(sequence_set_t* set, const char* path, size_t max_pairs) {
    FILE* f = fopen(path, "r");
    if (!f) { LOG_ERROR("Can not open %s", path) return false; }
    char *line = NULL, *pattern = NULL;
    size_t cap = 0, pcap = 0, plen = 0;
    ssize_t len;
    bool have_pattern = false, ok = true;
    while ((len = getline(&line, &cap, f)) >= 0) {
        const size_t n = chomp(line, len);
        if (n == 0) continue;
        if (line[0] == '>') {
            if (n > pcap) { pcap = n * 2; pattern = (char*)realloc(pattern, pcap); if (!pattern) { ok = false; break; } }
            plen = n - 1;
            memcpy(pattern, line + 1, plen);
            have_pattern = true;
        } else if (line[0] == '<') {
            if (!have_pattern) { LOG_ERROR("Malformed .seq file %s: text without pattern.", path) ok = false; break; }
            if (!set_append(set, pattern, plen, line + 1, n - 1)) { ok = false; break; }
            have_pattern = false;
            if (max_pairs && set->num_pairs >= max_pairs) break;
        } else {
            LOG_ERROR("Malformed .seq file %s: lines must start with '>' or '<'.", path)
            ok = false; break;
        }
    }
    free(line); free(pattern); fclose(f);
    return ok;
}

/* Reads the next FASTA record's residues (joined) into *seq; returns false at EOF. */
typedef struct { FILE* f; char* line; size_t cap; ssize_t pending; } fasta_t;

static bool fasta_next(fasta_t* fa, char** seq, size_t* seq_cap, size_t* seq_len) {
    *seq_len = 0;
    bool in_record = false;
    for (;;) {
        ssize_t len = fa->pending >= 0 ? fa->pending : getline(&fa->line, &fa->cap, fa->f);
        fa->pending = -1;
        if (len < 0) return in_record;
        const size_t n = chomp(fa->line, len);
        if (n == 0) continue;
        if (fa->line[0] == '>') {
            if (in_record) { fa->pending = (ssize_t)n; return true; }   /* header of the next record: keep it */
            in_record = true;
            continue;
        }
        if (!in_record) in_record = true;   /* headerless file */
        if (*seq_len + n + 1 > *seq_cap) {
            *seq_cap = (*seq_len + n + 1) * 2;
            *seq = (char*)realloc(*seq, *seq_cap);
            if (!*seq) return false;
        }
        memcpy(*seq + *seq_len, fa->line, n);
        *seq_len += n;
    }
}

bool read_fasta_pair_files(sequence_set_t* set, const char* query_path, const char* target_path, size_t max_pairs) {
    fasta_t q = {fopen(query_path, "r"), NULL, 0, -1}, t = {fopen(target_path, "r"), NULL, 0, -1};
    if (!q.f || !t.f) {
        LOG_ERROR("Can not open %s", !q.f ? query_path : target_path)
        if (q.f) fclose(q.f);
        if (t.f) fclose(t.f);
        return false;
    }
    char *qs = NULL, *ts = NULL;
    size_t qcap = 0, tcap = 0, qlen = 0, tlen = 0;
    bool ok = true;
    for (;;) {
        const bool hq = fasta_next(&q, &qs, &qcap, &qlen);
        const bool ht = fasta_next(&t, &ts, &tcap, &tlen);
        if (!hq || !ht) {
            if (hq != ht) LOG_WARN("Query and target files hold a different number of records; extra records ignored.")
            break;
        }
        if (!set_append(set, qs, qlen, ts, tlen)) { ok = false; break; }
        if (max_pairs && set->num_pairs >= max_pairs) break;
    }
    free(qs); free(ts); free(q.line); free(t.line); fclose(q.f); fclose(t.f);
    return ok;
}

void free_sequence_set(sequence_set_t* set) {
    free(set->sequences_buffer);
    free(set->sequences_metadata);
    memset(set, 0, sizeof(*set));
}
