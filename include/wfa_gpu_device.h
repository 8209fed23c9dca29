/*
 * wfa_gpu_device.h -- device-resident entry points of the MI355X build.
 *
 * The reference only exposes host-buffer entry points (lib/align.cuh:35-47):
 * every call re-uploads the ASCII sequences over PCIe.  These functions are
 * what launch_alignments* are built from, exported so that an integrator (or
 * bench.py) can keep a batch resident in HBM and so that parity tests can
 * exercise each stage (lib/sequence_packing.cu:96-116 pack,
 * lib/sequence_alignment.cu:211-470 align, utils/cigar.c:96-272 CIGAR
 * recovery) separately.  Plain C: device addresses travel as typed
 * pointers or void pointers, a HIP stream as a void pointer.
 */
#ifndef WFA_GPU_DEVICE_H
#define WFA_GPU_DEVICE_H

#include "wfa_gpu_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wfagpu_amd_ctx wfagpu_amd_ctx_t;

/* Tuning switches and test hooks of a context.  All zero = the defaults (what every product path uses); they are
 * plain configuration, set by the caller -- the library reads no environment variables. */
typedef struct {
    int min_tier;          /* test hook: skip the wavefront-kernel tiers below this one so that the rarely needed ones run
                              (1: 4 waves, 2: 16 waves, 3: ring in HBM, 4: hybrid ring)                                  */
    int careful_only;      /* diagnostics: keep every score on the careful (WFA2-to-the-letter) loop                     */
    int force_band;        /* with a band requested: always run the banded kernels (default: only where they pay)        */
    int no_auto_budget;    /* never tune per-pair score budgets from a sample                                            */
    int max_blocks_per_cu; /* occupancy experiments: cap on resident workgroups per CU (0: none)                         */
    int t0_min_blocks;     /* one-wave tier only while LDS leaves at least this many rings per CU (0: default, 10)       */
    int waves_per_simd;    /* one-wave exact kernels: force the instantiation compiled for 8, 7, 6 or 4 waves per SIMD
                              (0: matched to the rings LDS lets a CU hold)                                               */
    int trace_mode;        /* 0: automatic; 1: lane-per-alignment walk + windowed emit whatever the length (the fallback
                              of the wave-per-alignment kernel); 2: the wave-per-alignment kernel does the whole backtrace
                              (never several alignments per wavefront, never walk there + replay in the lane kernel); 3: never
                              the one-kernel backtrace of short alignments (walk + emit + compaction instead: A/B); 4: long
                              alignments walked by the wave-per-alignment kernel and replayed by the lane kernel whatever the
                              size of the pass (default: passes of 8192 alignments and more)                              */
    int timed_barriers;    /* diagnostics: the multi-wave exact CIGAR tiers (1 and 4) run an instantiation in which workgroup 0
                              records when each of its waves reaches and leaves the per-score barrier (s_memtime);
                              wfagpu_amd_debug_times() hands the records out                                              */
    int no_short_cigar;    /* A/B and test hook: CIGAR calls never take tier 5 (several alignments per wavefront)               */
    int no_fused_pack;     /* A/B: always run the pack kernel (default: reads of 512 bases and more -- and shorter ones where tier 5, several
                              alignments per wavefront, takes the penalties -- are packed by the wavefront kernels while they stage them) */
    int kernel_walk;       /* retired in round 6 (accepted, ignored): the one-wave wavefront kernels walking a finished alignment back
                              themselves -- measured a loss in round 5 (the backtrace pass 3.05 -> 1.45 ms, the wavefront kernel
                              25.4 -> 27.9 ms: EXPERIMENTS.md) -- went with the row table it read; the tile layout of the origin
                              bytes makes wfa_walk_kernel itself cheap                                                           */
    int exact_two_waves;   /* A/B hook: the exact search takes two waves per alignment where it would take four                 */
    int band_tier;         /* A/B hook: wavefronts per alignment of the banded kernels -- 1: one, 2: two, 3: four, 4: sixteen (0: by
                              the wavefronts a CU ends up holding, plan_tier)                                                    */
    int no_host_parts;     /* A/B and test hook: a score-only call whose only wavefront launch is tier 5 over the whole batch lets that
                              launch store its partial sums (cells, unfinished / failed / flagged pairs per wavefront) straight into
                              pinned host memory -- one kernel and a stream synchronisation per call --; 1: they stay on the device and
                              come over with the counter block (a copy behind the kernel)                                           */
    int short_iterations;  /* tier 5: iterations (groups of 4 / 2 alignments) a wavefront of a launch runs at least, where the list allows
                              it and four wavefronts per CU remain (0: default)                                                      */
    int arena_chunk_cap;   /* A/B: largest refill (16-byte units) a workgroup takes from the backtrace arena at a time (0: default, 262144 =
                              4 MiB; through round 4: 4096)                                                                        */
    int emit_pairs;        /* lane-per-alignment CIGAR replay with the sequences staged in LDS: alignments per wavefront
                              (8..64; 0: automatic -- as many as keep the most lanes resident per CU)                          */
    int verify_counters;   /* test hook: a call that starts WITHOUT zeroing the device's counter block -- because the call before it
                              claims to have left it zeroed (no kernel touched a counter, or a memset was queued behind the last read)
                              -- reads the block back first and fails if a word of it is not zero: the invariant that skipping the
                              memset rests on, checked instead of assumed                                            */
} wfagpu_amd_tuning_t;

typedef struct {
    int device;            /* HIP device ordinal                                         */
    void* stream;          /* hipStream_t to run on; NULL: the context creates its own (hipStreamNonBlocking: it is NOT ordered
                              behind the device's null stream -- see "Stream ordering" below); the null stream itself: null_stream */
    size_t arena_bytes;    /* backtrace arena (origin bytes + row headers); 0: automatic */
    size_t text_bytes;     /* CIGAR text arena; 0: automatic                             */
    size_t arena_limit_bytes; /* cap for the automatic arena size (0: none); a batch that needs
                              more runs in several passes                                   */
    size_t arena_limit_max_bytes; /* > arena_limit_bytes: the cap doubles, up to this value, after every call that
                              needed several passes (launch_alignments* no longer use it: a call that regrows its arena
                              stalls behind the driver's wipe of the memory it has just released) */
    wfagpu_amd_tuning_t tuning;
    int null_stream;       /* 1: run on the device's NULL (legacy default) stream -- the handle 0 cannot say so through `stream`,
                              where 0 means "create my own".  PyTorch's default stream is this one
                              (torch.cuda.current_stream().cuda_stream == 0 unless a torch.cuda.Stream is current). */
} wfagpu_amd_config_t;

/* Stream ordering.  Every kernel and copy of a context runs on ITS stream (wfagpu_amd_stream).  The device buffers a call is
 * given -- sequences, metadata, d_packed / d_flags / d_scores -- must be COMPLETE on that stream's timeline when the call is
 * made: produced on the same stream, or by work that has finished (synchronised) or that the context's stream has been made
 * to wait for (hipStreamWaitEvent).  A context that created its own stream is non-blocking: a fill or copy still queued on
 * another stream -- the null stream included -- races with the call's kernels (the reference copies synchronously around its
 * kernels: tests/test_packing_kernel.cu:225-306, lib/sequence_alignment.cu:87-147).  The calls themselves are blocking:
 * their results are complete when they return. */

typedef struct {
    const char* d_sequences;             /* device: ASCII buffer, reference layout       */
    size_t sequences_bytes;              /* size of the ASCII buffer (tier 5 indexes it by 32-bit dwords below 16 GiB; 0: judged by packed_bytes) */
    const sequence_pair_t* d_metadata;   /* device: packed offsets filled (see below)    */
    size_t num_pairs;
    size_t packed_bytes;                 /* value returned by wfagpu_amd_fill_packed_offsets */
    unsigned int max_seq_len;            /* longest pattern/text in the batch            */
    const void* d_packed;                /* device, optional: the packed_bytes of 2-bit words the metadata's packed offsets
                                            describe, already packed by the caller (wfagpu_host_pack_sequence; every byte
                                            of the batch must be one of ACGT).  Then d_sequences is not read (may be NULL)
                                            and the pack kernel does not run.  NULL: pack on the device.             */
} wfagpu_amd_batch_t;

typedef struct {
    /* per-kernel device time of the last call, milliseconds (HIP events on the
     * context's stream) */
    float pack_ms;
    float align_ms;        /* all align launches                                          */
    float trace_ms;
    float total_ms;        /* from the start of the call's first pack or wavefront kernel to the end of its last kernel (list
                              bookkeeping in front of the first wavefront launch -- a status memset, the length-bucket compaction,
                              the budget kernel -- lies outside it: microseconds)                                         */
    int align_launches;
    /* work accounting */
    unsigned long long cells;          /* wavefront cells computed (sum of widths)        */
    unsigned long long arena_units;    /* 16-byte units of backtrace arena used           */
    unsigned long long text_bytes;     /* CIGAR text bytes produced                       */
    unsigned int pairs_tier[6];        /* pairs finished per kernel tier (0: 1 wave, 1: 4 waves, 2: 16 waves, 3: HBM ring, 4: hybrid ring,
                                          5: short wavefronts -- several alignments per wave, score-only)                      */
    unsigned int pairs_retried;        /* pairs that needed a wider tier                  */
    unsigned int pairs_raw;            /* pairs with bytes outside ACGT (byte-compare kernels) */
    unsigned int pairs_banded;         /* pairs finished by the adaptive-band kernels          */
    unsigned int pairs_budget_missed;  /* pairs whose score exceeded the auto-tuned budget      */
    int auto_budget;                   /* largest auto-tuned score budget of the call (0: off)  */
    unsigned int sub_batches;          /* arena-bounded passes                            */
    size_t lds_bytes_tier0;
    int blocks_per_cu_tier0;
    /* the longest wavefront-kernel launch of the call (the "main launch": a call also has short
     * launches for the auto-budget sample and for re-runs) */
    float main_launch_ms;
    int main_launch_tier;
    unsigned int main_launch_pairs;
    unsigned long long main_launch_cells;
    unsigned long long main_launch_seq_bytes;   /* packed sequence bytes its pairs read */
    /* work of auto-budget samples whose pairs were aligned again with the batch (score-only launches on <= 4096 pairs):
     * not part of cells / align_launches / sub_batches / pairs_tier above */
    unsigned long long sample_cells;
    int sample_launches;
    unsigned int sample_passes;
    int waves_per_simd_tier0;          /* instantiation the first wavefront launch used (one-wave exact kernels: 8, 7, 6 or 4) */
    unsigned int pairs_walked_in_kernel; /* pairs of chains whose wavefront kernels walked their alignments back themselves (no wfa_walk_kernel) */
    unsigned int pairs_trace_split;    /* long alignments whose backtrace was walked by the wave-per-alignment kernel and replayed by
                                          the lane-per-alignment kernel (passes of 8192 and more; tuning.trace_mode 4: any)        */
} wfagpu_amd_stats_t;

/* 0 on success, negative on error (message on stderr). */
int wfagpu_amd_create(wfagpu_amd_ctx_t** ctx, const wfagpu_amd_config_t* cfg);
void wfagpu_amd_destroy(wfagpu_amd_ctx_t* ctx);

/* Device buffers a context outgrows DURING a call are kept aside instead of freed there (hipFree waits for the whole
 * device to go idle: with other contexts' kernels running it stalls the caller for milliseconds); they are freed by
 * wfagpu_amd_destroy, by a later call once they add up to 1 GiB, or here -- call it when the device is idle. */
void wfagpu_amd_trim(wfagpu_amd_ctx_t* ctx);

/* Loads the code objects of every kernel family on the context's device (one empty launch each on its stream; the
 * runtime loads a code object at the first launch of any of its kernels, 5-25 ms): a cold caller does this while it
 * waits for its first upload.  Returns without synchronising.  0 on success. */
int wfagpu_amd_prime(wfagpu_amd_ctx_t* ctx);

/* tuning.timed_barriers: the device buffer of the last call's barrier records -- per score of workgroup 0's alignments and
 * per wave three uint64: s_memtime at arrival, s_memtime at release, (score << 32 | wavefront width - 1) -- and the number
 * of waves per workgroup of the launch that wrote them; *records = scores recorded (0 when nothing was). */
void wfagpu_amd_debug_times(const wfagpu_amd_ctx_t* ctx, const void** d_records, unsigned int* records, int* waves);

/* The HIP stream (hipStream_t) the context runs on -- its own, or the one given at creation. */
void* wfagpu_amd_stream(const wfagpu_amd_ctx_t* ctx);

/* Replaces the tuning switches of a live context (between calls).  NULL: the defaults. */
void wfagpu_amd_set_tuning(wfagpu_amd_ctx_t* ctx, const wfagpu_amd_tuning_t* tuning);

/* Host helper: assigns text_offset_packed / pattern_offset_packed for n
 * records (what lib/align.cu:103-115 does inline) and returns the number of
 * packed bytes the batch needs.  Each sequence gets ceil(len/16)+1 words. */
size_t wfagpu_amd_fill_packed_offsets(sequence_pair_t* metadata, size_t n);

/* Host helper: packs src[0, len) into dst[0, ceil(len / 16)] -- the words the pack kernel writes for a sequence (code
 * (c & 6) >> 1, first base in the low bits, a zero word at the end); returns 1 when a byte outside ACGT was seen. */
int wfagpu_host_pack_sequence(const char* src, uint32_t len, uint32_t* dst);
int wfagpu_host_pack_sequence_scalar(const char* src, uint32_t len, uint32_t* dst);   /* (the portable path, for tests) */
/* A strip of n records of a batch: assigns their packed offsets from first_off on (what wfagpu_amd_fill_packed_offsets does,
 * strip by strip) and, with a staging buffer (the batch's packed words, NULL: offsets only), packs their sequences into it;
 * sequences_bytes: size of the caller's buffer (bytes that may be read).  Returns 1 when a byte outside ACGT was seen. */
int wfagpu_host_pack_strip(const char* sequences, size_t sequences_bytes, sequence_pair_t* metadata, size_t n,
                           size_t first_off, uint32_t* stage);

/* Stage 1 only: 2-bit packing.  d_packed must hold batch->packed_bytes;
 * d_flags (2 bytes per pair: pattern, text) receives 1 where a byte outside
 * ACGT was seen. */
int wfagpu_amd_pack_device(wfagpu_amd_ctx_t* ctx, const wfagpu_amd_batch_t* batch,
                           void* d_packed, unsigned char* d_flags);

/* The whole hot path on a resident batch: pack -> wavefront kernels (tier
 * escalation on the device, never on the CPU) -> backtrace + CIGAR text.
 *   band          BAND_NONE (-1) for the exact search, else the re-centring period of the adaptive
 *                 band heuristic and band_width its number of diagonals (the reference's -B and -t)
 *   d_scores      device int32[num_pairs], positive scores
 *   compute_cigar when true the CIGAR text stays in the context's arena:
 *                 *d_text, *d_off (uint64 byte offsets), *d_len (uint32 strlen)
 *                 are device pointers that stay valid THROUGH the next call on ctx (two sets of
 *                 buffers alternate: a pipelined caller downloads batch j under the kernels of j+1).
 * Blocking (returns after the stream has drained).  0 on success. */
int wfagpu_amd_align_device(wfagpu_amd_ctx_t* ctx, const wfagpu_amd_batch_t* batch,
                            affine_penalties_t penalties, int max_error, int band, int band_width,
                            bool compute_cigar, int32_t* d_scores,
                            const char** d_text, const unsigned long long** d_off,
                            const unsigned int** d_len);

void wfagpu_amd_last_stats(const wfagpu_amd_ctx_t* ctx, wfagpu_amd_stats_t* out);

/* on = 1: the following calls on ctx align batches drawn from the same stream of reads (the batches of one
 * launch_alignments call): the auto-tuned score budgets learnt from a sample of one batch are tried on the next ones
 * without sampling again (results stay exact: pairs that miss their budget are re-run, and a batch in which more than
 * 5 % do makes the next one sample again).  on = 0 forgets what was learnt. */
void wfagpu_amd_hint_same_stream(wfagpu_amd_ctx_t* ctx, int on);

/* How launch_alignments* run a call.  All zero = automatic.  Set between calls (not while one is running). */
typedef struct {
    int num_devices;          /* devices a call is sharded over (0: all visible)                                         */
    int virtual_devices;      /* tests: this many shards -- own threads, contexts, streams -- mapped round-robin onto the
                                 physical devices: exercises the multi-device path on one GPU                            */
    int lanes_per_device;     /* contexts per device working on alternate batches (0: 2 for big calls -- 3 when the
                                 sequences go up packed, see host_pack --, else 1)                                        */
    int batches_per_device;   /* a single huge batch is cut into this many so that the stages overlap (0: 16)            */
    size_t arena_limit_bytes; /* fixed backtrace-arena cap per lane (0: the lane's share of the device memory -- half of what
                                 was free, over all lanes --: an arena is sized once, by its first batch's expected need)      */
    size_t input_pool_bytes;  /* device memory for resident input per device (0: a quarter of the free memory, <= 24 GiB) */
    int numa_pin;             /* 0: pin a device's host threads to its NUMA node when several devices are used,
                                 1: always (test hook for one-GPU boxes), -1: never                                      */
    int timing;               /* 1: print the stage times of every device on stderr; 2: also a clock line per batch       */
    wfagpu_amd_tuning_t tuning;   /* handed to every context the calls create                                            */
    int host_pack;            /* 2-bit packing on the host, a quarter of the bytes over PCIe (the call is PCIe-bound when it is not
                                 GPU-bound).  0: when a device's share of the host threads is >= 4 and the call is big,
                                 1: always, -1: never (ASCII goes up, the pack kernel runs).  A batch holding a byte outside
                                 ACGT always goes up as ASCII                                                           */
    int host_pack_threads;    /* threads packing a batch (0: three quarters of a device's share of the host threads, 2..12)         */
    int ascii_every;          /* A/B hook, host_pack automatic, big calls: n >= 2: every n-th batch of a device's slice (the first one
                                 included) goes up as ASCII and is packed by the wavefront kernels while they stage it (measured a loss:
                                 profiles/r06/ab_ascii_every.txt); 0: off                                                             */
    int bring_up;             /* 0: the first device query of the process (get_num_cuda_devices, get_cuda_SM_count -- what the CLI
                                 and wfagpu_set_default_options call before any alignment) starts bringing the caller's
                                 CURRENT device up in a background thread: streams, lanes, code objects (all devices:
                                 wfagpu_amd_warmup); -1: never -- set it before the first query                           */
} wfagpu_amd_launch_config_t;

/* NULL: back to the defaults.  Changing `tuning` or `arena_limit_bytes` drops the cached per-device state. */
void wfagpu_amd_configure_launch(const wfagpu_amd_launch_config_t* cfg);

/* Stage times of the last launch_alignments* call, milliseconds of wall clock.  The per-stage figures are sums over
 * the batches of the busiest device (stages overlap: they do not add up to `total_ms`). */
typedef struct {
    double total_ms;        /* the whole call                                                         */
    double plan_ms;         /* sharding + batch planning before any device work                       */
    double acquire_ms;      /* contexts, streams, cached buffers (cold calls)                         */
    double prep_ms;         /* spans, packed offsets (host)                                           */
    double upload_ms;       /* H2D of sequences and metadata                                          */
    double upload_wait_ms;  /* uploader waiting for a free input slot                                 */
    double device_ms;       /* wfagpu_amd_align_device, all lanes                                     */
    double device_wait_ms;  /* lanes waiting for their batch to arrive                                */
    double d2h_ms;          /* results to pinned staging                                              */
    double scatter_ms;      /* staging -> the caller's records                                        */
    double check_ms;        /* -c                                                                     */
    int devices, lanes, batches;
    unsigned host_threads;  /* host cores the call could use (affinity mask capped by the cgroup CPU quota) */
    double host_pack_ms;    /* 2-bit packing on the host (part of prep_ms)                             */
    int host_packed_batches;/* batches that went up packed                                            */
    int host_pack_threads;
} wfagpu_amd_launch_stats_t;
void wfagpu_amd_last_launch_stats(wfagpu_amd_launch_stats_t* out);
/* The same per device slot of the last call (shard = 0 .. devices-1): `devices` then holds the physical device of the slot,
 * `host_threads` the slot's share of the host threads, `total_ms` the slot's own wall.  0, or -1 when there is no such slot. */
int wfagpu_amd_last_launch_stats_device(int shard, wfagpu_amd_launch_stats_t* out);

/* Shorthand for wfagpu_amd_launch_config_t::num_devices. */
void wfagpu_amd_set_num_devices(int n);

/* Pairs that failed the check_correctness (-c) verification in the last launch_alignments* call (the "Incorrect=" counts
 * of its batch lines, summed).  The reference only prints them (lib/align.cu:318-326); the CLI turns a non-zero value
 * into a non-zero exit code. */
long wfagpu_amd_check_failures(void);

/* Starts bringing up, in the background, every device a call would be sharded over (upload/download streams, three lanes,
 * code objects: what a cold launch_alignments* call otherwise does under its first batches, ~100 ms): returns at once, a
 * call that follows waits only for what is still missing; the caller's current HIP device is left as it was.  The first
 * device query of the process (get_num_cuda_devices, get_cuda_SM_count) does the same for ONE device -- the caller's current
 * one, device 0 unless the process selected another -- unless wfagpu_amd_launch_config_t::bring_up is -1: a process that
 * will call launch_alignments* over several devices calls this function itself (the CLI does). */
void wfagpu_amd_warmup(void);

/* Blocks until every bring-up that has been started (the first device query, wfagpu_amd_warmup) is through.  The reference's
 * CUDA context exists when its device queries return (tools/aligner.c:189-204), before launch_alignments* is timed; a caller
 * that wants its timed call to start from the same state calls this first (the CLI does, in front of its "Wall time" clock: a
 * 2 GB input is read in less time than a cold device takes to come up). */
void wfagpu_amd_warmup_wait(void);

/* launch_alignments* keep their per-device state (context, backtrace arena,
 * input buffers, pinned result staging) for the next call of the process:
 * allocating it is most of the cost of a cold call.  This frees it. */
void wfagpu_amd_release_cache(void);

#ifdef __cplusplus
}
#endif
#endif
