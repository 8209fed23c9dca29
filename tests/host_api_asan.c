/* AddressSanitizer + UBSan harness for the HOST C code of the library and its tools (built and run by tests/test_sanitizers.py;
 * CPU only -- the GPU pool has no sanitizers):
 *   lib/aligner.c, lib/alignment_results.c      the wfagpu_* aligner object: growth paths past 1 MiB / 10 000 records, empty and
 *                                               32 767-base sequences, rejected arguments, wfagpu_align through a stub launcher
 *   utils/sequence_reader.c                     .seq and paired FASTA parsers on well-formed, CRLF, headerless, truncated,
 *                                               unequal and empty files, -n below / at / past the end of the file
 *   utils/verification.c                        CIGAR checkers on valid, truncated, overlong and garbage CIGARs; the scorer
 *   tools/generate_dataset.c                    both generators at the edges (length 0 and 1, buffers too small)
 * The seam symbols that live in csrc/wfa_launch.hip (launch_alignments*, device queries) are stubs here: the launcher scores
 * every pair with utils/verification.c's scalar scorer and writes a one-run CIGAR through the same buffer discipline the real
 * one uses (realloc when the caller's buffer is too small). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../include/wfa_gpu_abi.h"
#include "../wfa-gpu_amd/utils/sequence_reader.h"
#include "../wfa-gpu_amd/utils/verification.h"

size_t wfagen_pair_stride(int length, double error);
size_t wfagen_generate(char* seqbuf, size_t cap, sequence_pair_t* meta, size_t n, int length, double error, uint64_t seed, int nthreads);

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)

/* ---- stubs for the seam (csrc/wfa_launch.hip) ---- */
static long g_launches = 0;
static void stub_launch(char* seq, size_t seq_bytes, sequence_pair_t* meta, wfa_alignment_result_t* res, wfa_alignment_options_t opt, bool cigar) {
    if (!seq || !meta || !res) return;
    ++g_launches;
    for (size_t i = 0; i < opt.num_alignments; ++i) {
        const sequence_pair_t* m = &meta[i];
        CHECK(m->pattern_offset + m->pattern_len <= seq_bytes && m->text_offset + m->text_len <= seq_bytes);
        res[i].error = (unsigned)verification_cpu_score(seq + m->pattern_offset, seq + m->text_offset, m->pattern_len, m->text_len,
                                                        opt.penalties.x, opt.penalties.o, opt.penalties.e);
        if (cigar) {
            char tmp[32];
            const int n = snprintf(tmp, sizeof tmp, "%uM", m->pattern_len < m->text_len ? m->pattern_len : m->text_len);
            if ((size_t)n + 1 > res[i].cigar.buffer_size || !res[i].cigar.buffer) {
                res[i].cigar.buffer = realloc(res[i].cigar.buffer, (size_t)n + 1);
                res[i].cigar.buffer_size = (size_t)n + 1;
            }
            memcpy(res[i].cigar.buffer, tmp, (size_t)n + 1);
            res[i].cigar.last_free_position = (size_t)n;
        }
    }
}
void launch_alignments(char* s, const size_t sb, sequence_pair_t* const m, wfa_alignment_result_t* const r, wfa_alignment_options_t o, bool c) { (void)c; stub_launch(s, sb, m, r, o, true); }
void launch_alignments_distance(char* s, const size_t sb, sequence_pair_t* const m, wfa_alignment_result_t* const r, wfa_alignment_options_t o, bool c) { (void)c; stub_launch(s, sb, m, r, o, false); }
void get_num_cuda_devices(int* n) { *n = 1; }
int get_cuda_SM_count(int dev) { (void)dev; return 256; }
char* get_cuda_dev_name(int dev) { (void)dev; return strdup("stub"); }
void get_cuda_capability(int dev, int* a, int* b) { (void)dev; *a = 9; *b = 5; }

static uint32_t rnd_state = 4242;
static uint32_t rnd(void) { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }
static char* random_seq(size_t len) {
    char* s = malloc(len + 1);
    for (size_t i = 0; i < len; ++i) s[i] = "ACGT"[rnd() & 3];
    s[len] = 0;
    return s;
}

static void aligner_object(void) {
    wfagpu_aligner_t al;
    CHECK(!wfagpu_initialize_aligner(NULL));
    CHECK(wfagpu_initialize_aligner(&al));
    CHECK(!wfagpu_add_sequences(NULL, "A", "A") && !wfagpu_add_sequences(&al, NULL, "A") && !wfagpu_add_sequences(&al, "A", NULL));
    /* 12 000 records (the metadata array grows past its first 10 000), a few MiB of sequences (the buffer grows in 1 MiB
     * steps), empty sequences, the longest sequence the API accepts and one base more */
    char* longest = random_seq(32767);
    char* too_long = random_seq(32768);
    CHECK(!wfagpu_add_sequences(&al, too_long, "A") && !wfagpu_add_sequences(&al, "A", too_long));
    size_t added = 0;
    for (int i = 0; i < 12000; ++i) {
        /* (mostly short: the stub launcher scores every pair with the scalar checker, quadratic in the score) */
        const size_t top = (i % 200 == 0) ? 600 : 60;
        const size_t pl = (i % 97 == 0) ? 0 : rnd() % top, tl = (i % 89 == 0) ? 0 : rnd() % top;
        char* p = random_seq(pl); char* t = random_seq(tl);
        CHECK(wfagpu_add_sequences(&al, p, t));
        free(p); free(t);
        ++added;
    }
    CHECK(wfagpu_add_sequences(&al, longest, longest)); ++added;
    CHECK(wfagpu_add_sequences(&al, "", "")); ++added;
    CHECK(al.num_sequence_pairs == added);
    for (size_t i = 0; i < added; ++i) {
        const sequence_pair_t* m = &al.sequences_metadata[i];
        CHECK(m->pattern_offset % 4 == 0 && m->text_offset % 4 == 0);
        CHECK(m->pattern_offset + m->pattern_len < m->text_offset || m->pattern_len == 0);
        CHECK(al.sequences_buffer[m->pattern_offset + m->pattern_len] == 0 && al.sequences_buffer[m->text_offset + m->text_len] == 0);
    }
    affine_penalties_t pen = {2, 3, 1}, bad = {-1, 3, 1}, zero = {0, 0, 0};
    CHECK(!wfagpu_initialize_parameters(NULL, pen) && !wfagpu_initialize_parameters(&al, bad) && !wfagpu_initialize_parameters(&al, zero));
    CHECK(wfagpu_initialize_parameters(&al, pen));
    CHECK(!wfagpu_set_batch_size(NULL, 10));
    CHECK(wfagpu_set_batch_size(&al, 0) && wfagpu_set_batch_size(&al, added * 2) && wfagpu_set_batch_size(&al, 1000));
    CHECK(!wfagpu_align(NULL));
    al.alignment_options.compute_cigar = false;
    CHECK(wfagpu_align(&al));
    al.alignment_options.compute_cigar = true;
    CHECK(wfagpu_align(&al));
    CHECK(g_launches == 2);
    CHECK(al.results[added - 1].error == 0 && al.results[added - 2].error == 0);      /* identical / empty pairs */
    for (size_t i = 0; i < added; ++i) CHECK(al.results[i].cigar.buffer && strlen(al.results[i].cigar.buffer) == al.results[i].cigar.last_free_position);
    /* more sequences after an alignment (the results array belongs to the old count: the object re-initialises it) */
    CHECK(wfagpu_add_sequences(&al, "ACGT", "ACGA"));
    CHECK(wfagpu_initialize_parameters(&al, pen));
    CHECK(wfagpu_align(&al) && al.results[added].error == 2);
    wfagpu_destroy_aligner(&al);
    wfagpu_destroy_aligner(&al);      /* (twice: every pointer is reset) */
    wfagpu_destroy_aligner(NULL);
    free(longest); free(too_long);
    /* results arrays on their own */
    wfa_alignment_result_t* r = NULL;
    CHECK(!initialize_wfa_results(NULL, 4, 16));
    CHECK(initialize_wfa_results(&r, 0, 0) && destroy_wfa_results(r, 0));
    CHECK(initialize_wfa_results(&r, 1000, 0) && destroy_wfa_results(r, 1000));
    CHECK(!destroy_wfa_results(NULL, 3));
}

static const char* tmp_dir;
static char path_buf[4][512];
static const char* write_file(int slot, const char* name, const char* content, size_t len) {
    snprintf(path_buf[slot], sizeof path_buf[slot], "%s/%s", tmp_dir, name);
    FILE* f = fopen(path_buf[slot], "wb");
    CHECK(f != NULL);
    if (len) CHECK(fwrite(content, 1, len, f) == len);
    fclose(f);
    return path_buf[slot];
}
static void check_set(const sequence_set_t* s) {
    for (size_t i = 0; i < s->num_pairs; ++i) {
        const sequence_pair_t* m = &s->sequences_metadata[i];
        CHECK(m->pattern_offset % 4 == 0 && m->text_offset % 4 == 0);
        CHECK(m->pattern_offset + m->pattern_len < s->sequences_buffer_used + 1 && m->text_offset + m->text_len < s->sequences_buffer_used + 1);
        CHECK(s->sequences_buffer[m->pattern_offset + m->pattern_len] == 0 && s->sequences_buffer[m->text_offset + m->text_len] == 0);
    }
}
static void readers(void) {
    sequence_set_t s;
    /* a well-formed .seq file that makes both arrays grow, with and without the final newline, LF and CRLF */
    size_t cap = 8u << 20, len = 0;
    char* big = malloc(cap);
    for (int i = 0; i < 11000; ++i) {
        const size_t pl = (i % 113 == 0) ? 0 : rnd() % 500, tl = (i % 127 == 0) ? 0 : rnd() % 500;
        big[len++] = '>'; for (size_t j = 0; j < pl; ++j) big[len++] = "ACGT"[rnd() & 3]; big[len++] = '\n';
        big[len++] = '<'; for (size_t j = 0; j < tl; ++j) big[len++] = "ACGT"[rnd() & 3]; big[len++] = '\n';
    }
    for (int variant = 0; variant < 3; ++variant) {
        const char* p = write_file(0, "big.seq", big, variant == 1 ? len - 1 : len);
        memset(&s, 0, sizeof s);
        CHECK(read_seq_file(&s, p, variant == 2 ? 7 : 0));
        CHECK(s.num_pairs == (variant == 2 ? 7u : 11000u));
        check_set(&s);
        free_sequence_set(&s);
    }
    memset(&s, 0, sizeof s);
    CHECK(read_seq_file(&s, path_buf[0], 1000000) && s.num_pairs == 11000);      /* -n past the end of the file */
    /* inputs that cannot be mapped -- a FIFO, /dev/stdin, `-i <(zcat x.seq.gz)`: st_size 0 -- go through the getline reader: the same
     * records and bytes as the mapped read of the same file, with and without -n (such inputs came back as zero pairs) */
    {
        char fifo[4200];
        snprintf(fifo, sizeof fifo, "%s.fifo", path_buf[0]);
        unlink(fifo);
        CHECK(mkfifo(fifo, 0600) == 0);
        for (int round = 0; round < 2; ++round) {
            const size_t limit = round ? 1234 : 0;
            const pid_t pid = fork();
            CHECK(pid >= 0);
            if (pid == 0) {      /* the writer: the file into the pipe (a reader that stops early, -n, just closes it) */
                signal(SIGPIPE, SIG_IGN);
                FILE* in = fopen(path_buf[0], "rb"); FILE* out = fopen(fifo, "wb");
                if (!in || !out) _exit(2);
                char chunk[65536]; size_t got;
                while ((got = fread(chunk, 1, sizeof chunk, in)) > 0) if (fwrite(chunk, 1, got, out) != got) break;
                fclose(in); fclose(out);
                _exit(0);
            }
            sequence_set_t piped;
            memset(&piped, 0, sizeof piped);
            CHECK(read_seq_file(&piped, fifo, limit));
            int status = 0;
            CHECK(waitpid(pid, &status, 0) == pid);
            sequence_set_t mapped;
            memset(&mapped, 0, sizeof mapped);
            CHECK(read_seq_file(&mapped, path_buf[0], limit));
            CHECK(piped.num_pairs == (limit ? limit : 11000u) && piped.num_pairs == mapped.num_pairs);
            check_set(&piped);
            for (size_t i = 0; i < piped.num_pairs; ++i) {
                const sequence_pair_t *a = &piped.sequences_metadata[i], *b = &mapped.sequences_metadata[i];
                CHECK(a->pattern_len == b->pattern_len && a->text_len == b->text_len && a->has_N == b->has_N);
                CHECK(memcmp(piped.sequences_buffer + a->pattern_offset, mapped.sequences_buffer + b->pattern_offset, a->pattern_len) == 0);
                CHECK(memcmp(piped.sequences_buffer + a->text_offset, mapped.sequences_buffer + b->text_offset, a->text_len) == 0);
            }
            free_sequence_set(&piped); free_sequence_set(&mapped);
        }
        unlink(fifo);
    }
    /* the file is parsed in strips by several threads (sequence_reader.c): the same records and the same bytes from 1, 2, 3, 7
     * and 32 strips, with -n cutting inside a strip, at a strip's first pair and beyond the file; lines with CR LF, blank lines
     * and a doubled pattern line (the last one counts) across the cuts */
    {
        size_t len2 = 0;
        char* mixed = malloc(cap + (1u << 20));
        for (int i = 0; i < 9000; ++i) {
            const size_t pl = rnd() % 300, tl = rnd() % 300;
            if (i % 97 == 0) { mixed[len2++] = '>'; mixed[len2++] = 'T'; mixed[len2++] = 'T'; mixed[len2++] = '\n'; }      /* overwritten by the next '>' line */
            mixed[len2++] = '>'; for (size_t j = 0; j < pl; ++j) mixed[len2++] = "ACGTN"[rnd() % 5 == 4 && rnd() % 50 == 0 ? 4 : rnd() & 3];
            if (i % 3 == 0) mixed[len2++] = '\r';
            mixed[len2++] = '\n';
            if (i % 41 == 0) mixed[len2++] = '\n';
            mixed[len2++] = '<'; for (size_t j = 0; j < tl; ++j) mixed[len2++] = "ACGT"[rnd() & 3];
            if (i % 3 == 0) mixed[len2++] = '\r';
            mixed[len2++] = '\n';
        }
        const char* mp = write_file(1, "mixed.seq", mixed, len2);
        static const size_t limits[] = {0, 1, 2999, 3000, 8999, 9000, 20000};
        for (size_t li = 0; li < sizeof limits / sizeof limits[0]; ++li) {
            sequence_set_t ref;
            memset(&ref, 0, sizeof ref);
            sequence_reader_force_threads = 1;
            CHECK(read_seq_file(&ref, mp, limits[li]));
            CHECK(ref.num_pairs == (limits[li] && limits[li] < 9000 ? limits[li] : 9000u));
            check_set(&ref);
            static const int strips[] = {2, 3, 7, 32};
            for (size_t si = 0; si < sizeof strips / sizeof strips[0]; ++si) {
                sequence_set_t got;
                memset(&got, 0, sizeof got);
                sequence_reader_force_threads = strips[si];
                CHECK(read_seq_file(&got, mp, limits[li]));
                CHECK(got.num_pairs == ref.num_pairs && got.sequences_buffer_used == ref.sequences_buffer_used);
                CHECK(memcmp(got.sequences_metadata, ref.sequences_metadata, ref.num_pairs * sizeof(sequence_pair_t)) == 0);
                CHECK(memcmp(got.sequences_buffer, ref.sequences_buffer, ref.sequences_buffer_used) == 0);
                free_sequence_set(&got);
            }
            free_sequence_set(&ref);
        }
        /* a malformed line deep inside the file is found whichever strip it falls into */
        mixed[len2 / 2] = '\n'; mixed[len2 / 2 + 1] = 'A';
        mp = write_file(1, "mixed_bad.seq", mixed, len2);
        for (int t = 1; t <= 7; t += 3) {
            sequence_set_t got;
            memset(&got, 0, sizeof got);
            sequence_reader_force_threads = t;
            CHECK(!read_seq_file(&got, mp, 0));
            free_sequence_set(&got);
            memset(&got, 0, sizeof got);
            CHECK(read_seq_file(&got, mp, 100) && got.num_pairs == 100);      /* (-n stops before it, like the reference's reader) */
            free_sequence_set(&got);
        }
        sequence_reader_force_threads = 0;
        free(mixed);
    }
    free_sequence_set(&s);
    free(big);
    static const char crlf[] = ">ACGT\r\n<ACGA\r\n>AC\r\n<\r\n";
    memset(&s, 0, sizeof s);
    CHECK(read_seq_file(&s, write_file(0, "crlf.seq", crlf, sizeof crlf - 1), 0) && s.num_pairs == 2);
    CHECK(s.sequences_metadata[0].pattern_len == 4 && s.sequences_metadata[1].text_len == 0);
    free_sequence_set(&s);
    /* malformed: wrong order, a pattern without its text, a line without marker, no file, an empty file */
    static const char* bad[] = {"<ACGT\n>ACGT\n", ">ACGT\n", ">ACGT\nACGT\n", ">ACGT\n<ACGT\n>ACGT\n", "ACGT\n"};
    for (size_t i = 0; i < sizeof bad / sizeof bad[0]; ++i) {
        memset(&s, 0, sizeof s);
        const bool ok = read_seq_file(&s, write_file(0, "bad.seq", bad[i], strlen(bad[i])), 0);
        CHECK(!ok || s.num_pairs <= 1);
        free_sequence_set(&s);
    }
    memset(&s, 0, sizeof s);
    CHECK(!read_seq_file(&s, "/nonexistent/dir/x.seq", 0));
    free_sequence_set(&s);
    memset(&s, 0, sizeof s);
    (void)read_seq_file(&s, write_file(0, "empty.seq", "", 0), 0);
    CHECK(s.num_pairs == 0);
    free_sequence_set(&s);
    /* paired FASTA: multi-line records, CRLF, blank lines, unequal record counts, a file without any header, -n */
    static const char q[] = ">q0 first\nACGTAC\nGTAC\n\n>q1\nAC\r\nGT\r\n>q2\n>q3\nTTTT";
    static const char t[] = ">t0\nACGTACGTAC\n>t1\nACGT\n>t2\nA\n";
    const char* qp = write_file(1, "q.fasta", q, sizeof q - 1);
    const char* tp = write_file(2, "t.fasta", t, sizeof t - 1);
    memset(&s, 0, sizeof s);
    CHECK(read_fasta_pair_files(&s, qp, tp, 0));
    CHECK(s.num_pairs == 3 && s.sequences_metadata[0].pattern_len == 10 && s.sequences_metadata[1].pattern_len == 4 && s.sequences_metadata[2].pattern_len == 0);
    check_set(&s);
    free_sequence_set(&s);
    memset(&s, 0, sizeof s);
    CHECK(read_fasta_pair_files(&s, qp, tp, 2) && s.num_pairs == 2);
    free_sequence_set(&s);
    memset(&s, 0, sizeof s);
    static const char headerless[] = "ACGTACGT\nACGT\n";
    (void)read_fasta_pair_files(&s, write_file(3, "h.fasta", headerless, sizeof headerless - 1), tp, 0);
    check_set(&s);
    free_sequence_set(&s);
    memset(&s, 0, sizeof s);
    CHECK(!read_fasta_pair_files(&s, "/nonexistent/q.fa", tp, 0));
    free_sequence_set(&s);
    /* a 3 MiB single-line record (the line buffers grow): refused, the readers take sequences below MAX_SEQ_LEN like
     * wfagpu_add_sequences; the longest one they take: 32 767 bases */
    const size_t L = 3u << 20;
    char* rec = malloc(L + 8);
    rec[0] = '>'; rec[1] = 'x'; rec[2] = '\n';
    for (size_t i = 0; i < L; ++i) rec[3 + i] = "ACGT"[i & 3];
    rec[3 + L] = '\n';
    const char* lp = write_file(3, "long.fasta", rec, L + 4);
    memset(&s, 0, sizeof s);
    CHECK(!read_fasta_pair_files(&s, lp, lp, 0) && s.num_pairs == 0);
    free_sequence_set(&s);
    rec[3 + 32767] = '\n';
    lp = write_file(3, "longest.fasta", rec, 3 + 32767 + 1);
    memset(&s, 0, sizeof s);
    CHECK(read_fasta_pair_files(&s, lp, lp, 0) && s.num_pairs == 1 && s.sequences_metadata[0].pattern_len == 32767);
    check_set(&s);
    free_sequence_set(&s);
    rec[0] = '>'; rec[3 + 32767] = '\n';
    memmove(rec + 1, rec + 3, 32767); rec[1 + 32767] = '\n'; rec[2 + 32767] = '<'; memcpy(rec + 3 + 32767, rec + 1, 100); rec[3 + 32767 + 100] = '\n';
    memset(&s, 0, sizeof s);
    CHECK(read_seq_file(&s, write_file(3, "longest.seq", rec, 3 + 32767 + 101), 0) && s.num_pairs == 1 && s.sequences_metadata[0].pattern_len == 32767);
    free_sequence_set(&s);
    free(rec);
}

static void checkers(void) {
    const char* p = "ACGTACGTAC";      /* pattern */
    const char* t = "ACGAACGTTTAC";    /* text    */
    CHECK(check_cigar_edit(t, p, 12, 10, "3M1X4M2I2M"));
    CHECK(check_affine_distance(t, p, 12, 10, 2 + 3 + 2, 2, 3, 1, "3M1X4M2I2M"));
    CHECK(!check_affine_distance(t, p, 12, 10, 6, 2, 3, 1, "3M1X4M2I2M"));
    static const char* garbage[] = {"", "M", "3", "3M1X4M2I", "3M1X4M2I2M5M", "99999999999999999999M", "3M1Q4M", "-3M", "3M1X4M2I2", "0M", "12I10D", "10D12I",
                                    "3M1X4M2D2M", "4294967295M", "3M 1X", "3m1x"};
    for (size_t i = 0; i < sizeof garbage / sizeof garbage[0]; ++i) {
        (void)check_cigar_edit(t, p, 12, 10, garbage[i]);
        (void)check_affine_distance(t, p, 12, 10, 7, 2, 3, 1, garbage[i]);
        (void)check_cigar_edit(t, p, 0, 0, garbage[i]);
    }
    CHECK(!check_cigar_edit(t, p, 12, 10, "3M1X4M2I2M5M") && !check_cigar_edit(t, p, 12, 10, "3M1X4M2I") && !check_cigar_edit(t, p, 12, 10, ""));
    CHECK(check_cigar_edit("", "", 0, 0, ""));
    verification_scratch_t scratch = {NULL, 0};
    for (int i = 0; i < 300; ++i) {
        const size_t pl = rnd() % 300, tl = rnd() % 300;
        char* a = random_seq(pl); char* b = random_seq(tl);
        const int s1 = verification_cpu_score_scratch(a, b, pl, tl, 2, 3, 1, &scratch);
        const int s2 = verification_cpu_score(a, b, pl, tl, 2, 3, 1);
        CHECK(s1 == s2 && s1 >= 0);
        CHECK(verification_cpu_score_scratch(a, a, pl, pl, 5, 3, 2, &scratch) == 0);
        free(a); free(b);
    }
    CHECK(verification_cpu_score_scratch("", "ACGT", 0, 4, 2, 3, 1, &scratch) == 3 + 4);
    CHECK(verification_cpu_score_scratch("ACGT", "", 4, 0, 2, 3, 1, &scratch) == 3 + 4);
    verification_scratch_free(&scratch);
    verification_scratch_free(&scratch);
}

static void generator(void) {
    for (int length = 0; length <= 3; ++length) {
        const size_t n = 50, stride = wfagen_pair_stride(length, 0.5);
        const size_t cap = stride * n + 16;
        char* buf = malloc(cap);
        sequence_pair_t* meta = calloc(n, sizeof *meta);
        CHECK(wfagen_generate(buf, cap - 1, meta, n, length, 0.5, 7, 2) == 0);      /* one byte short: refused, nothing written */
        CHECK(wfagen_generate(buf, cap, meta, n, length, 0.5, 7, 2) != 0);
        for (size_t i = 0; i < n; ++i) CHECK(meta[i].text_len == (unsigned)length && meta[i].pattern_offset + meta[i].pattern_len < cap && meta[i].text_offset + meta[i].text_len < cap);
        free(buf); free(meta);
    }
    const size_t n = 200, stride = wfagen_pair_stride(1000, 0.05), cap = stride * n + 16;
    char* buf = malloc(cap);
    sequence_pair_t* meta = calloc(n, sizeof *meta);
    CHECK(wfagen_generate(buf, cap, meta, n, 1000, 0.05, 9, 4) != 0);
    for (size_t i = 0; i < n; ++i) {
        CHECK(buf[meta[i].pattern_offset + meta[i].pattern_len] == 0 && buf[meta[i].text_offset + meta[i].text_len] == 0);
        for (unsigned j = 0; j < meta[i].pattern_len; ++j) CHECK(strchr("ACGT", buf[meta[i].pattern_offset + j]) != NULL);
    }
    free(buf); free(meta);
}

int main(int argc, char** argv) {
    tmp_dir = argc > 1 ? argv[1] : "/tmp";
    aligner_object();
    readers();
    checkers();
    generator();
    printf("host_api_asan ok\n");
    return 0;
}
