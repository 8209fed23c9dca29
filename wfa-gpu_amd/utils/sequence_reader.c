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

/* ---- .seq files: parsed by all the cores the process may use ------------------------------------------------------------
 * The reference's reader (utils/sequence_reader.c:137-227) is one getline loop; on a 1M-pair file (2 GB) that loop IS the run
 * time of the tool: 2.4 s of a 2.6 s process around a 0.06-0.12 s alignment call.  Here the file is mapped, cut at the
 * starts of '>' lines into one strip per thread, and read in two sweeps: the first counts each strip's pairs and the bytes
 * they take in the padded batch layout, a prefix sum gives every strip its place, the second copies the sequences and fills
 * the records -- same layout, same error cases (a '<' line without a '>' line before it, a line that starts with neither, a
 * sequence that is too long), blank lines skipped, CR LF accepted, -n honoured. */
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

int sequence_reader_force_threads = 0;      /* test hook: strips of a .seq file whatever its size (0: by size and cores) */

static unsigned reader_threads(size_t bytes) {
    if (sequence_reader_force_threads > 0) return sequence_reader_force_threads > 32 ? 32u : (unsigned)sequence_reader_force_threads;
    unsigned n = 1;
    cpu_set_t cs;
    if (sched_getaffinity(0, sizeof cs, &cs) == 0) n = (unsigned)CPU_COUNT(&cs);
    FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");      /* (an affinity mask of 256 CPUs over a quota of 16 cores) */
    if (f) {
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long cores = atol(q) / period;
            if (cores >= 1 && (unsigned)cores < n) n = (unsigned)cores;
        }
        fclose(f);
    }
    if (n > 32) n = 32;
    const size_t by_size = bytes / ((size_t)8 << 20) + 1;      /* (a strip of less than 8 MB is not worth a thread) */
    if (n > by_size) n = (unsigned)by_size;
    return n ? n : 1;
}

typedef struct {
    const char* base; size_t from, to;      /* this strip: lines that START in [from, to) */
    size_t pairs, bytes;                    /* sweep 1: pairs of the strip, bytes of the layout they take */
    size_t first_pair, first_byte, take;    /* sweep 2: where they go, how many of them are wanted (-n) */
    sequence_set_t* set;
    int err;                                /* 1: text without pattern, 2: bad line start, 3: sequence too long */
    bool fill;
} strip_t;

static void* strip_sweep(void* arg) {
    strip_t* st = (strip_t*)arg;
    const char* const base = st->base;
    size_t pos = st->from, pairs = 0, bytes = 0;
    const char* pat = NULL; size_t plen = 0;
    char* const out = st->fill ? st->set->sequences_buffer : NULL;
    size_t off = st->first_byte;
    while (pos < st->to) {
        const char* nl = (const char*)memchr(base + pos, '\n', st->to - pos);
        /* (the last line of the file may lack its newline; a strip never ends inside a line: see the cuts) */
        size_t end = nl ? (size_t)(nl - base) : st->to;
        const size_t next = nl ? end + 1 : st->to;
        while (end > pos && (base[end - 1] == '\r' || base[end - 1] == '\n')) --end;
        const size_t n = end - pos;
        if (n == 0) { pos = next; continue; }
        const char c = base[pos];
        if (c == '>') { pat = base + pos + 1; plen = n - 1; }
        else if (c == '<') {
            if (!pat) { st->err = 1; st->pairs = pairs; st->bytes = bytes; return NULL; }
            const size_t tlen = n - 1;
            if (plen >= MAX_SEQ_LEN || tlen >= MAX_SEQ_LEN) { st->err = 3; st->pairs = pairs; st->bytes = bytes; return NULL; }
            const size_t psz = WFA_ALIGN_32_BITS(plen + 1), tsz = WFA_ALIGN_32_BITS(tlen + 1);
            if (st->fill) {
                if (pairs >= st->take) break;
                const size_t poff = off, toff = off + psz;
                memcpy(out + poff, pat, plen); memset(out + poff + plen, 0, psz - plen);
                memcpy(out + toff, base + pos + 1, tlen); memset(out + toff + tlen, 0, tsz - tlen);
                sequence_pair_t* m = &st->set->sequences_metadata[st->first_pair + pairs];
                memset(m, 0, sizeof *m);
                m->pattern_offset = poff; m->pattern_len = (unsigned int)plen;
                m->text_offset = toff; m->text_len = (unsigned int)tlen;
                m->has_N = memchr(pat, 'N', plen) != NULL || memchr(base + pos + 1, 'N', tlen) != NULL;
                off += psz + tsz;
            }
            bytes += psz + tsz;
            ++pairs;
            pat = NULL;
        } else { st->err = 2; st->pairs = pairs; st->bytes = bytes; return NULL; }
        pos = next;
    }
    st->pairs = pairs; st->bytes = bytes;
    return NULL;
}

static void run_strips(strip_t* st, unsigned n, bool fill) {
    pthread_t th[32];
    bool started[32] = {false};
    for (unsigned i = 0; i < n; ++i) { st[i].fill = fill; st[i].err = 0; }
    for (unsigned i = 1; i < n; ++i) started[i] = pthread_create(&th[i], NULL, strip_sweep, &st[i]) == 0;
    strip_sweep(&st[0]);
    for (unsigned i = 1; i < n; ++i) { if (started[i]) pthread_join(th[i], NULL); else strip_sweep(&st[i]); }
}

/* The reference's reader as it is (utils/sequence_reader.c:137-227: one getline loop): for inputs that cannot be mapped -- a FIFO,
 * /dev/stdin, a process substitution (`-i <(zcat x.seq.gz)`): st_size says 0 for those -- and when mmap fails.  Same layout, same
 * error cases, blank lines skipped, CR LF accepted, -n honoured. */
static bool read_seq_stream(sequence_set_t* set, int fd, const char* path, size_t max_pairs) {
    FILE* f = fdopen(fd, "r");
    if (!f) { LOG_ERROR("Can not open %s", path) close(fd); return false; }
    char *line = NULL, *pat = NULL;
    size_t cap = 0, pat_cap = 0, plen = 0;
    bool have_pat = false, ok = true;
    ssize_t len;
    while ((len = getline(&line, &cap, f)) >= 0) {
        const size_t n = chomp(line, len);
        if (n == 0) continue;
        if (line[0] == '>') {
            if (n > pat_cap) { char* np = (char*)realloc(pat, n + 1); if (!np) { LOG_ERROR("Out of memory reading %s", path) ok = false; break; } pat = np; pat_cap = n; }
            memcpy(pat, line + 1, n - 1); plen = n - 1; have_pat = true;
        } else if (line[0] == '<') {
            if (!have_pat) { LOG_ERROR("Malformed .seq file %s: text without pattern.", path) ok = false; break; }
            if (!set_append(set, pat, plen, line + 1, n - 1)) { ok = false; break; }
            have_pat = false;
            if (max_pairs && set->num_pairs >= max_pairs) break;
        } else { LOG_ERROR("Malformed .seq file %s: lines must start with '>' or '<'.", path) ok = false; break; }
    }
    free(line); free(pat); fclose(f);
    return ok;
}

bool read_seq_file(sequence_set_t* set, const char* path, size_t max_pairs) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { LOG_ERROR("Can not open %s", path) return false; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { LOG_ERROR("Can not stat %s", path) close(fd); return false; }
    /* (only a regular file has a size to map and to cut into strips) */
    if (!S_ISREG(sb.st_mode)) return read_seq_stream(set, fd, path, max_pairs);
    const size_t size = (size_t)sb.st_size;
    if (size == 0) { close(fd); return true; }
    const char* base = (const char*)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) return read_seq_stream(set, fd, path, max_pairs);
    close(fd);
    (void)madvise((void*)base, size, MADV_WILLNEED);
    /* (with -n only the head of the file is looked at: grown until it holds the pairs asked for) */
    unsigned nt = reader_threads(size);
    strip_t st[32];
    bool ok = true;
    size_t span = size;
    if (max_pairs) { span = (size_t)1 << 22; if (span > size) span = size; }
    for (;;) {
        if (max_pairs && span < size) {
            /* end the head at a line start */
            const char* nl = (const char*)memchr(base + span - 1, '\n', size - (span - 1));
            span = nl ? (size_t)(nl - base) + 1 : size;
        }
        const unsigned n_now = (max_pairs && span < ((size_t)64 << 20) && sequence_reader_force_threads <= 0) ? 1u : nt;
        /* cuts: strip i starts at the first '>' line that starts at or after i * span / n (strip 0 at the start of the file) */
        size_t cut[33];
        cut[0] = 0; cut[n_now] = span;
        for (unsigned i = 1; i < n_now; ++i) {
            size_t p = span / n_now * i;
            if (p < cut[i - 1]) p = cut[i - 1];
            /* the next line start, then on to a line that starts with '>' */
            while (p < span) {
                const char* nl = p ? (const char*)memchr(base + p - 1, '\n', span - (p - 1)) : base - 1;
                if (!nl) { p = span; break; }
                p = (size_t)(nl - base) + 1;
                if (p >= span || base[p] == '>') break;
                ++p;
            }
            cut[i] = p;
        }
        for (unsigned i = 0; i < n_now; ++i) { st[i] = (strip_t){0}; st[i].base = base; st[i].from = cut[i]; st[i].to = cut[i + 1]; st[i].set = set; }
        run_strips(st, n_now, false);
        size_t pairs = 0, bytes = 0, pairs_before_err = 0;
        int err = 0;
        for (unsigned i = 0; i < n_now; ++i) {
            if (st[i].err && !err) { err = st[i].err; pairs_before_err = pairs + st[i].pairs; }
            st[i].first_pair = pairs; st[i].first_byte = bytes;
            pairs += st[i].pairs; bytes += st[i].bytes;
        }
        /* (with -n the reference's reader stops at the last pair it was asked for: what follows is never looked at) */
        if (err && !(max_pairs && pairs_before_err >= max_pairs)) {
            if (err == 1) LOG_ERROR("Malformed .seq file %s: text without pattern.", path)
            else if (err == 2) LOG_ERROR("Malformed .seq file %s: lines must start with '>' or '<'.", path)
            else LOG_ERROR("Sequence %zu is too long (max %lu bases).", pairs_before_err, MAX_SEQ_LEN - 1)
            ok = false; break;
        }
        if (max_pairs && pairs < max_pairs && span < size) { span = span * 4 < size ? span * 4 : size; continue; }      /* a longer head */
        if (max_pairs && pairs > max_pairs) pairs = max_pairs;
        /* places: every strip's pairs that are wanted */
        size_t total_bytes = 0;
        {
            size_t left = pairs;
            for (unsigned i = 0; i < n_now; ++i) { st[i].take = st[i].pairs < left ? st[i].pairs : left; left -= st[i].take; }
            /* (bytes of a truncated strip: known after the fill; an upper bound sizes the buffer) */
            for (unsigned i = 0; i < n_now; ++i) if (st[i].take) total_bytes = st[i].first_byte + st[i].bytes;
        }
        /* (a large buffer on huge pages where the kernel hands them out on request: 2 GB of sequences are 500k first-touch faults in
         * the strips' threads and 0.2 s of unmapping at the end on 4 KiB pages; still a free()-able block) */
        set->sequences_buffer = NULL;
        if (total_bytes + 64 >= ((size_t)32 << 20)) {
            void* big = NULL;
            if (posix_memalign(&big, (size_t)2 << 20, total_bytes + 64) == 0) {
                (void)madvise(big, total_bytes + 64, MADV_HUGEPAGE);
                set->sequences_buffer = (char*)big;
            }
        }
        if (!set->sequences_buffer) set->sequences_buffer = (char*)malloc(total_bytes + 64);
        set->sequences_metadata = (sequence_pair_t*)malloc((pairs ? pairs : 1) * sizeof(sequence_pair_t));
        if (!set->sequences_buffer || !set->sequences_metadata) {
            LOG_ERROR("Out of memory reading %s", path)
            free(set->sequences_buffer); free(set->sequences_metadata);      /* (whichever of the two was allocated) */
            set->sequences_buffer = NULL; set->sequences_metadata = NULL;
            ok = false; break;
        }
        set->sequences_buffer_size = total_bytes + 64;
        set->metadata_capacity = pairs ? pairs : 1;
        run_strips(st, n_now, true);
        size_t used = 0;
        for (unsigned i = 0; i < n_now; ++i) if (st[i].take) used = st[i].first_byte + st[i].bytes;      /* (bytes: of the pairs filled) */
        memset(set->sequences_buffer + used, 0, total_bytes + 64 - used);
        set->sequences_buffer_used = used;
        set->num_pairs = pairs;
        break;
    }
    munmap((void*)base, size);
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
