/*
 * wfa.affine.gpu -- command line of the MI355X build.  Same options, input
 * formats, defaults and output lines as the reference CLI
 * (tools/aligner.c:58-517): it reads the whole input, derives -e/-t/-w/-b
 * defaults the same way, calls launch_alignments / launch_alignments_distance
 * directly, prints "Alignment computed. Wall time: ...", and writes
 * "score<TAB>CIGAR" lines (score negative, as WFA2's align_benchmark does).
 */
#include <getopt.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/wfa_gpu_device.h"
#include "../utils/logger.h"
#include "../utils/sequence_reader.h"
#include "../utils/wf_clock.h"

/* The first device query of a process initialises the HIP runtime (0.2-0.3 s) and starts bringing the device up in the background
 * (wfa_launch.hip): it runs on a thread of its own while the main thread parses the options and reads the input -- a 2 GB .seq file is
 * 0.25 s --, and is joined before anything asks about the device. */
static int g_ndev = 0;
static double g_query_seconds = 0.0;
static void* device_query_thread(void* arg) {
    (void)arg;
    const double t0 = wf_now_seconds();
    get_num_cuda_devices(&g_ndev);
    /* (no device: said at once -- the main thread may be in the middle of a multi-GB read that nothing will use) */
    if (g_ndev == 0) { LOG_ERROR("No HIP devices detected.") exit(-1); }
    if (g_ndev > 0) {
        int major = 0, minor = 0;
        get_cuda_capability(0, &major, &minor);
        char* name = get_cuda_dev_name(0);
        LOG_INFO("Using HIP device \"%s\" with capability %d.%d (%d visible)", name, major, minor, g_ndev)
        free(name);
        /* (the query above started bringing device 0 up in the background; the call will be sharded over every visible device) */
        if (g_ndev > 1) wfagpu_amd_warmup();
    }
    g_query_seconds = wf_now_seconds() - t0;
    return NULL;
}

static void usage(const char* prog) {
    printf("Options:\n[Input/Output]\n"
           "\t-i, --input-seq <file>            Sequences to align in .seq format\n"
           "\t-Q, --input-fasta-query <file>    Query sequences in FASTA format (with -T)\n"
           "\t-T, --input-fasta-target <file>   Target sequences in FASTA format (with -Q)\n"
           "\t-n, --num-alignments <int>        Number of alignments to read (default: all)\n"
           "\t-o, --output-file <file>          Write 'score<TAB>CIGAR' lines to file\n"
           "\t-p, --print-output                Print the output to stderr instead\n"
           "\t-O, --output-verbose              Add the sequences to every output line\n"
           "[Alignment]\n"
           "\t-g, --affine-penalties <x,o,e>    Gap-affine penalties (default 2,3,1)\n"
           "\t-x, --compute-cigar               Compute the alignment path (CIGAR), not just the score\n"
           "\t-e, --max-distance <int>          Expected maximum score (sizes the first kernel tier)\n"
           "\t-b, --batch-size <int>            Alignments per batch (default: all)\n"
           "\t-B, --band <int|auto>             Adaptive band: re-centring period (auto = 25)\n"
           "\t-c, --check                       Verify results against a CPU computation\n"
           "[System]\n"
           "\t-t, --threads-per-block <int>     Accepted for compatibility (band width in banded mode)\n"
           "\t-w, --workers <int>               Accepted for compatibility\n"
           "\t    --stage-times                 Print the stage clocks of the alignment call on stderr\n"
           "[Examples]\n"
           "\t%s -i sequences.seq -b <batch_size> -o scores.out\n"
           "\t%s -i sequences.seq -b <batch_size> -B auto -o scores-banded.out\n"
           "\t%s -Q queries.fasta -T targets.fasta -b <batch_size> -o scores.out\n"
           "\t%s -Q queries.fasta -T targets.fasta -b <batch_size> -x -o cigars.out\n",
           prog, prog, prog, prog);
}

int main(int argc, char** argv) {
    static const struct option longopts[] = {
        {"input-seq", required_argument, 0, 'i'}, {"input-fasta-query", required_argument, 0, 'Q'},
        {"input-fasta-target", required_argument, 0, 'T'}, {"num-alignments", required_argument, 0, 'n'},
        {"output-file", required_argument, 0, 'o'}, {"print-output", no_argument, 0, 'p'},
        {"output-verbose", no_argument, 0, 'O'}, {"affine-penalties", required_argument, 0, 'g'},
        {"compute-cigar", no_argument, 0, 'x'}, {"max-distance", required_argument, 0, 'e'},
        {"batch-size", required_argument, 0, 'b'}, {"band", required_argument, 0, 'B'},
        {"check", no_argument, 0, 'c'}, {"threads-per-block", required_argument, 0, 't'},
        {"workers", required_argument, 0, 'w'}, {"help", no_argument, 0, 'h'}, {"stage-times", no_argument, 0, 1000}, {0, 0, 0, 0}};
    const char *seq_path = NULL, *q_path = NULL, *t_path = NULL, *out_path = NULL, *pen_str = NULL;
    long n_read = 0, max_distance = -1, batch_size = -1, band_arg = -2, tpb = -1, workers = -1;
    bool print_out = false, verbose = false, cigar = false, check = false, stage_times = false;


    int c;
    opterr = 0;      /* (unknown options are skipped silently: see the default case) */
    while ((c = getopt_long(argc, argv, "i:Q:T:n:o:pOg:xe:b:B:ct:w:h", longopts, NULL)) != -1) {
        switch (c) {
            case 'i': seq_path = optarg; break;
            case 'Q': q_path = optarg; break;
            case 'T': t_path = optarg; break;
            case 'n': n_read = atol(optarg); break;
            case 'o': out_path = optarg; break;
            case 'p': print_out = true; break;
            case 'O': verbose = true; break;
            case 'g': pen_str = optarg; break;
            case 'x': cigar = true; break;
            case 'e': max_distance = atol(optarg); if (max_distance <= 0) { LOG_ERROR("Maximum error supported by the kernel must be > 0. Aborting.") exit(-1); } break;
            case 'b': batch_size = atol(optarg); break;
            case 'B': band_arg = atol(optarg); break;   /* "auto" parses as 0 (utils/arg_handler.c:54) */
            case 'c': check = true; break;
            case 't': tpb = atol(optarg); break;
            case 'w': workers = atol(optarg); break;
            case 1000: stage_times = true; break;      /* (not in the reference: the library's stage clocks of the call on stderr) */
            case 'h': usage(argv[0]); exit(0);
            default: break;      /* an option the tool does not know is skipped, like the reference's parser does (utils/arg_handler.c:97-140) */
        }
    }
    if (!seq_path && !(q_path && t_path)) {
        LOG_ERROR("No input file provided.")
        usage(argv[0]);
        return 1;
    }

    int x = 2, o = 3, e = 1;
    if (pen_str && sscanf(pen_str, "%d,%d,%d", &x, &o, &e) != 3) {
        LOG_WARN("Invalid penalties format provided. Using default penalties (0,2,3,1).")
        x = 2; o = 3; e = 1;
    }
    if (x < 0) x = -x;
    if (o < 0) o = -o;
    if (e < 0) e = -e;
    const affine_penalties_t penalties = {x, o, e};
    LOG_INFO("Penalties: M=0, X=%d, O=%d, E=%d.", x, o, e)

    /* (started here, behind the option checks -- no exit path of the tool leaves it running --, joined right behind the read) */
    pthread_t query_thread;
    const bool query_started = pthread_create(&query_thread, NULL, device_query_thread, NULL) == 0;
    if (!query_started) device_query_thread(NULL);
    LOG_INFO("Reading sequences file...")
    sequence_set_t set;
    memset(&set, 0, sizeof set);
    const double t_start = wf_now_seconds();
    double t0 = t_start;
    const bool ok = seq_path ? read_seq_file(&set, seq_path, (size_t)(n_read > 0 ? n_read : 0))
                             : read_fasta_pair_files(&set, q_path, t_path, (size_t)(n_read > 0 ? n_read : 0));
    const double t_read = wf_now_seconds() - t0;
    if (query_started) pthread_join(query_thread, NULL);
    const double t_query_wait = wf_now_seconds() - t0 - t_read;
    if (!ok) { LOG_ERROR("Error reading input.") exit(1); }
    if (set.num_pairs == 0) { LOG_ERROR("No sequence pairs found in the input.") exit(1); }
    LOG_INFO("File read: %.3fs (%zu pairs)", t_read, set.num_pairs)

    if (g_ndev == 0) { LOG_ERROR("No HIP devices detected.") exit(-1); }

    if (max_distance < 0) {   /* tools/aligner.c:319-338 */
        max_distance = (long)(MAX(set.sequences_metadata[0].text_len, set.sequences_metadata[0].pattern_len) * 0.1);
        max_distance *= MAX(x, MAX(o, e));
        if (max_distance <= 20) max_distance = 20;
        LOG_INFO("No maximum error provided by the user, using %ld", max_distance)
    }
    if (tpb < 0) tpb = wfa_get_threads_per_alignment((size_t)max_distance);
    const size_t num_alignments = set.num_pairs;
    if (batch_size < 0) batch_size = (long)num_alignments;
    if (batch_size <= 0) { LOG_ERROR("Incorrect batch size (%ld).", batch_size) exit(-1); }
    LOG_INFO("Batch size = %ld.", batch_size)
    if (workers < 0) workers = get_num_workers((int)tpb);
    if (workers <= 0) { LOG_ERROR("Incorrect number of workers (%ld).", workers) exit(-1); }
    int band = BAND_NONE;
    if (band_arg != -2) {
        if (band_arg < 0) { LOG_ERROR("Band must positive (band=%ld).", band_arg) exit(-1); }
        band = band_arg == 0 ? 25 : (int)band_arg;
        LOG_INFO("Banded execution. Band width: %ld. Band re-centering every %d steps", tpb, band)
    }

    wfa_alignment_result_t* results = NULL;
    const double t_res0 = wf_now_seconds();
    /* (the reference starts every CIGAR buffer at max_distance * 5 bytes: 1.5 GB of calloc for a million 1 kbp pairs, 0.4 s of the
     * tool's run time.  An RLE CIGAR of score s has at most s operations, each at most ~5 characters with its match run; the typical
     * one is a quarter of that bound, and a buffer that is too small grows when the results are scattered) */
    if (!initialize_wfa_results(&results, num_alignments, cigar ? (size_t)max_distance * 5 / 4 + 16 : 1)) {
        LOG_ERROR("Can not initialise CIGAR buffer.")
        exit(-1);
    }
    const double t_results = wf_now_seconds() - t_res0;
    wfa_alignment_options_t opt;
    memset(&opt, 0, sizeof opt);
    opt.max_error = (int)max_distance; opt.threads_per_block = (int)tpb; opt.num_workers = (int)workers;
    opt.band = band; opt.batch_size = (size_t)batch_size; opt.num_alignments = num_alignments;
    opt.penalties = penalties; opt.compute_cigar = cigar;

    if (stage_times) {
        wfagpu_amd_launch_config_t lc;
        memset(&lc, 0, sizeof lc);
        lc.timing = 2;
        wfagpu_amd_configure_launch(&lc);
    }
    /* (the device started coming up with the device query, in the background: the clock below times the call from the state the
     * reference's call starts in -- context there) */
    const double t_wait0 = wf_now_seconds();
    wfagpu_amd_warmup_wait();
    const double t_bring_wait = wf_now_seconds() - t_wait0;
    t0 = wf_now_seconds();
    if (cigar) launch_alignments(set.sequences_buffer, set.sequences_buffer_size, set.sequences_metadata, results, opt, check);
    else launch_alignments_distance(set.sequences_buffer, set.sequences_buffer_size, set.sequences_metadata, results, opt, check);
    const double secs = wf_now_seconds() - t0;
    printf("Alignment computed. Wall time: %.3fs (%.3f alignments per second)\n", secs, (double)num_alignments / secs);
    /* (the reference's clock covers the allocations and module load its launch_alignments does, tools/aligner.c:450-474 -- its CUDA
     * context exists by then; here the device's background bring-up (streams, lanes, code objects) is waited for in front of the clock:
     * what that wait was is said next to the number it is not part of) */
    printf("Device bring-up waited for before the clock: %.3fs (wall time including it: %.3fs)\n", t_bring_wait, secs + t_bring_wait);

    const double t_out0 = wf_now_seconds();
    if (out_path || print_out) {
        FILE* fp = stderr;
        if (!print_out) {
            LOG_INFO("Writing output file...")
            fp = fopen(out_path, "w");
            if (!fp) { LOG_ERROR("Could not open file %s", out_path) exit(-1); }
        }
        for (size_t i = 0; i < num_alignments; ++i) {
            const char* cg = cigar ? results[i].cigar.buffer : "";
            if (verbose) {
                const sequence_pair_t* m = &set.sequences_metadata[i];
                fprintf(fp, "%d\t%s\t%.*s\t%.*s\n", -(int)results[i].error, cg, (int)m->pattern_len,
                        set.sequences_buffer + m->pattern_offset, (int)m->text_len, set.sequences_buffer + m->text_offset);
            } else {
                fprintf(fp, "%d\t%s\n", -(int)results[i].error, cg);
            }
        }
        if (!print_out) fclose(fp);
    }
    const double t_output = wf_now_seconds() - t_out0, t_free0 = wf_now_seconds();
    destroy_wfa_results(results, num_alignments);
    const double t_free_results = wf_now_seconds() - t_free0;
    free_sequence_set(&set);
    (void)t_free_results;
    if (stage_times)
        fprintf(stderr, "[cli stages] read %.3f s, device query %.3f on a thread of its own (%.3f of it behind the read), results array %.3f, alignment call %.3f, "
                        "output %.3f, release %.3f; %.3f waiting for the device's bring-up in front of the call; %.3f since the options were parsed\n",
                t_read, g_query_seconds, t_query_wait, t_results, secs, t_output, wf_now_seconds() - t_free0, t_bring_wait, wf_now_seconds() - t_start);
    if (check && wfagpu_amd_check_failures() > 0) {
        LOG_ERROR("%ld alignments failed the -c verification.", wfagpu_amd_check_failures())
        return 2;
    }
    return 0;
}
