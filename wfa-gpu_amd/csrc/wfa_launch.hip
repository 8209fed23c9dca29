// The C-ABI seam: launch_alignments / launch_alignments_distance
// (lib/align.cuh:35-47 of the reference, implemented there by
// lib/align.cu:42-881) and the device-query shims
// (utils/device_query.cu:27-54).
//
// Host buffers in, host results out, blocking, results in input order.  The
// pairs of a call are sharded in contiguous slices of equal work over the
// visible GPUs (no collective: pairs are independent); a device's slice is cut
// into batches of options.batch_size like the reference does and runs as a
// pipeline whose stages never wait for each other's buffers:
//
//   prep thread      spans + packed offsets of every batch (host only, runs ahead) and -- when the host has the cores --
//                    the 2-bit packing of the batch into a staging ring (utils/host_pack.c: the pack kernel's words,
//                    30 GB/s of ASCII per core), so that a quarter of the bytes cross PCIe
//   upload thread    ONE stream of back-to-back H2D copies into a pool of input slots -- with 288 GB of HBM the
//                    whole slice stays resident (1M x 1 kbp pairs: 2 GB of ASCII, 0.5 GB packed), so the copy engine never
//                    idles while a batch computes: the call is bound by PCIe (ASCII: 2 GB at ~56 GB/s = 36 ms) or by
//                    the kernels, whichever is longer, plus the tail of the last batch
//   K compute lanes  contexts (stream, backtrace arena, scratch) that each take the next batch nobody has taken: the host
//                    round trips and backtrace kernels of one lane are filled by the wavefront kernels of the others
//   K result lanes   D2H of a batch's results into pinned staging -- under the kernels of the lane's NEXT batch: a context
//                    alternates between two sets of output buffers --, then staging -> the caller's
//                    wfa_alignment_result_t records (+ the -c check)
//
// The reference overlaps the same phases by hand with two streams and double buffers (lib/align.cu:63-68,177-385).
// The library reads no environment variables: see wfagpu_amd_launch_config_t.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include "../../include/wfa_gpu_device.h"
#include "../utils/logger.h"
#include "../utils/verification.h"

namespace {

constexpr int MAX_DEV = 64;     // device slots (physical devices, or virtual ones in tests)
constexpr int MAX_LANES = 4;    // compute lanes per device

std::mutex g_cfg_mu;
wfagpu_amd_launch_config_t g_cfg{};
wfagpu_amd_launch_stats_t g_last_stats{};
std::vector<wfagpu_amd_launch_stats_t> g_last_shard_stats;      // (per device slot of the last call; host_threads: the slot's share)
std::atomic<long> g_check_failures{0};   // pairs that failed the -c check in the last launch_alignments* call

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container with 16 cores'
// worth of quota on a 256-thread host sees 256 in its mask; 256 busy threads would only be throttled).  The budget the
// devices of a call share for their scatter and -c workers.
unsigned usable_host_threads() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = (unsigned)c; }
  auto cap = [&](long long quota, long long period) {
    if (quota > 0 && period > 0) n = (unsigned)std::max<long long>(1, std::min<long long>(n, (quota + period / 2) / period));
  };
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota|max> <period>"
    char q[32] = {0};
    long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) cap(atoll(q), period);
    fclose(f);
  } else {
    long long quota = -1, period = 0;                               // cgroup v1
    if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
    if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
    cap(quota, period);
  }
  return n;
}

// fn(t) for t in [0, nt) on nt threads (the caller's included)
void parallel_for(unsigned nt, const std::function<void(unsigned)>& fn) {
  if (nt <= 1) { fn(0); return; }
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(fn, t);
  fn(0);
  for (auto& t : pool) t.join();
}

// Best effort: run a device's host threads (this one and the ones it starts: they inherit the mask) on the cores of the
// NUMA node its GPU hangs off, so that the staging copies and the scatter of eight devices do not all cross the socket
// interconnect.  Intersected with the mask the process already has; any failure leaves things as
// they are.
void pin_to_device_node(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf) - 1, device) != hipSuccess) return;
  for (char* p = bdf; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');    // sysfs spells it in lower case
  char path[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return;
  int node = -1;
  const int got = fscanf(f, "%d", &node);
  fclose(f);
  if (got != 1 || node < 0) return;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  f = fopen(path, "r");
  if (!f) return;
  char list[4096] = {0};
  const size_t len = fread(list, 1, sizeof(list) - 1, f);
  fclose(f);
  if (!len) return;
  cpu_set_t have, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof(have), &have) != 0) return;
  for (char* p = list; *p;) {                     // "0-31,64-95"
    char* end = nullptr;
    long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); if (end == p) break; }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET((int)c, &have)) CPU_SET((int)c, &want);
    p = (*end == ',') ? end + 1 : end;
    if (*end != ',') break;
  }
  if (CPU_COUNT(&want) > 0) (void)sched_setaffinity(0, sizeof(want), &want);
}

struct Shard {
  int device;        // physical device
  int slot;          // index of the cached per-device state (== device unless virtual devices are configured)
  int sharers = 1;   // slots that share this physical device (their arenas share its memory)
  size_t from, to;   // [from, to)
  unsigned host_threads = 1;   // this device's share of the host cores (scatter and -c workers): SURVEY.md section 8(e)
  int rc = 0;
  wfagpu_amd_launch_stats_t st{};
};

struct CallArgs {
  char* seq;
  size_t seq_bytes;
  sequence_pair_t* meta;
  wfa_alignment_result_t* results;
  wfa_alignment_options_t opt;
  bool check;
  bool cigar;
  wfagpu_amd_launch_config_t cfg;
};

#define HIP_OK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      LOG_ERROR("HIP call %s failed: %s", #expr, hipGetErrorString(_e));                     \
      return -1;                                                                             \
    }                                                                                        \
  } while (0)

int check_batch(const CallArgs& a, size_t from, size_t to, int batch_idx, unsigned max_threads) {
  // lib/align.cu:258-326 (CIGAR mode) / :688-739 (distance mode)
  std::atomic<long> correct{0}, incorrect{0};
  std::atomic<long long> sum{0};
  std::atomic<long> next{(long)from};
  auto worker = [&](unsigned) {
    verification_scratch_t scratch = {nullptr, 0};     // this worker's, for all its pairs; freed below
    for (;;) {
      const long i0 = next.fetch_add(16);
      if (i0 >= (long)to) break;
      const long i1 = std::min<long>(i0 + 16, (long)to);
      for (long i = i0; i < i1; ++i) {
        const sequence_pair_t& m = a.meta[i];
        const char* text = a.seq + m.text_offset;
        const char* pattern = a.seq + m.pattern_offset;
        const int dist = (int)a.results[i].error;
        bool ok = true;
        if (a.cigar) {
          const char* cg = a.results[i].cigar.buffer;
          const bool c1 = check_cigar_edit(text, pattern, m.text_len, m.pattern_len, cg);
          const bool c2 = check_affine_distance(text, pattern, m.text_len, m.pattern_len, dist, a.opt.penalties.x,
                                                a.opt.penalties.o, a.opt.penalties.e, cg);
          if (!c1) LOG_ERROR("Incorrect cigar (%ld). Distance: %d. CIGAR: %s", i, dist, cg);
          ok = c1 && c2;
        }
        const int cpu = verification_cpu_score_scratch(pattern, text, m.pattern_len, m.text_len, a.opt.penalties.x,
                                                       a.opt.penalties.o, a.opt.penalties.e, &scratch);
        if (cpu != dist) { LOG_ERROR("Incorrect distance (%ld). GPU=%d, CPU=%d", i, dist, cpu); ok = false; }
        sum += dist;
        if (ok) ++correct; else ++incorrect;
      }
    }
    verification_scratch_free(&scratch);
  };
  // (each device has its share of the host cores: with 8 devices a pool of every hardware thread per batch and lane
  // would oversubscribe the host many times over and starve the upload and scatter threads)
  unsigned nt = std::max(1u, std::min(64u, max_threads));
  nt = (unsigned)std::min<size_t>(nt, (to - from + 15) / 16);
  parallel_for(nt, worker);
  fprintf(stderr, "(Batch %d) correct=%ld Incorrect=%ld Average score=%f\n", batch_idx, correct.load(), incorrect.load(),
          (to > from) ? (double)sum.load() / (double)(to - from) : 0.0);
  g_check_failures += incorrect.load();
  return incorrect.load() ? 1 : 0;
}

// Per-device state that outlives a call: the compute lanes (context = stream + backtrace arena + scratch, score buffer,
// two sets of pinned result staging each), the pool of input slots and the upload stream.  Allocating them is most of
// the cost of a cold call (pinned host memory ~0.25 ms per MiB, device memory ~33 ms per GiB at first touch), so they
// are kept for the next call of the process and freed by wfagpu_amd_release_cache() or at exit by the OS.  Calls are
// single-threaded by contract (lib/aligner.h of the reference is not re-entrant); a mutex per slot guards the cache anyway.
struct Lane {
  wfagpu_amd_ctx_t* ctx = nullptr;
  int32_t* d_scores[2] = {nullptr, nullptr}; size_t scores_cap[2] = {0, 0};      // (alternate like the context's CIGAR buffers)
  struct Out {
    char* text = nullptr; size_t text_cap = 0;
    unsigned long long* off = nullptr; unsigned int* len = nullptr; size_t cig_cap = 0;   // CIGAR calls only
    int32_t* score = nullptr; size_t n_cap = 0;
  } out[2];
};
struct InSlot { char* d_seq = nullptr; size_t seq_cap = 0; sequence_pair_t* d_meta = nullptr; size_t meta_cap = 0;
                uint32_t* d_packed = nullptr; size_t packed_cap = 0; };
struct HostStage { uint32_t* p = nullptr; size_t cap = 0; };      // host-packed words of a batch on their way up (pageable)
constexpr int STAGE_RING = 4;
struct DevState {
  int device = -1;
  // H2D of the batches, in order.  ONE stream: the runtime moves pageable memory in 32 MiB pieces with ~30 us of host
  // work between them (53.7 GB/s instead of the 57 GB/s of a piece); feeding two streams from two threads closes
  // those gaps but costs more than it gives -- whole batches on alternate streams arrive in pairs, twice as late (1M x
  // 1 kbp pairs: 51 ms instead of 43), two halves of every batch on two streams: upload 38.6 -> 37.2 ms, call 43.2 -> 45.0
  // (five busy host threads on a 16-core quota).
  hipStream_t up = nullptr;
  hipStream_t down = nullptr;                  // D2H of every lane's results (one stream: the copies share one PCIe direction anyway, and
                                               // every stream of a new priority level is a hardware queue to create: ~11 ms, serialised)
  std::vector<hipEvent_t> up_done, down_done;  // one per batch of a call: "its copies have landed"
  std::vector<InSlot> in;
  void* d_scratch = nullptr; void* d_scratch2 = nullptr; void* h_scratch = nullptr;      // (a few KiB each: the two ends of the copies that warm the copy paths up: one device buffer per direction)
  // ---- bring-up (see bring_up_fn): who creates what, in which order
  std::mutex bring_mu;                         // publishes items (up, down, lane[k].ctx) and bring_active
  std::mutex create_mu;                        // one creator at a time; never held while waiting for bring_mu's condition
  std::condition_variable bring_cv;
  std::thread bring_thread;
  bool bring_active = false;                   // (under bring_mu)
  std::atomic<int> prime_state{0};             // 0: code objects not loaded, 1: somebody is at it, 2: done
  int sharers = 1;                             // slots sharing the physical device when the lanes were created (their arena caps are shares)
  double bring_clock[MAX_LANES + 3] = {0};     // (timing: when the bring-up thread had the upload stream, lane k, the download stream: now_ms())
  HostStage stage[STAGE_RING];
  Lane lane[MAX_LANES];
  wfagpu_amd_tuning_t tuning{};                // what the contexts were created with
  size_t arena_limit_cfg = 0;
  size_t free_at_creation = 0, total_mem = 0;  // device memory when the slot was set up (the lanes' arena caps are shares of it)
};
DevState g_dev[MAX_DEV];
std::mutex g_dev_mu[MAX_DEV];   // one per slot: the devices of a call run concurrently

void join_bring(DevState& d) { if (d.bring_thread.joinable()) d.bring_thread.join(); }
// (a process that ends while a device is still coming up -- or after a call that failed half-way -- must neither tear the
// runtime down under that thread nor destroy a joinable std::thread: registered by whoever starts the first bring-up thread)
void join_all_bring_at_exit() {
  static std::once_flag at_exit;
  std::call_once(at_exit, [] { atexit([] { for (int i = 0; i < MAX_DEV; ++i) join_bring(g_dev[i]); }); });
}

void release_dev(DevState& d) {
  join_bring(d);
  if (d.device < 0) return;
  (void)hipSetDevice(d.device);
  // (the warm-up copies of a bring-up nobody has used yet may still be in flight on the copy streams)
  if (d.up) (void)hipStreamSynchronize(d.up);
  if (d.down) (void)hipStreamSynchronize(d.down);
  for (auto& in : d.in) { if (in.d_seq) (void)hipFree(in.d_seq); if (in.d_meta) (void)hipFree(in.d_meta); if (in.d_packed) (void)hipFree(in.d_packed); }
  d.in.clear();
  for (auto& hs : d.stage) { free(hs.p); hs = HostStage{}; }
  for (auto& e : d.up_done) (void)hipEventDestroy(e);
  d.up_done.clear();
  for (auto& e : d.down_done) (void)hipEventDestroy(e);
  d.down_done.clear();
  if (d.d_scratch) (void)hipFree(d.d_scratch);
  if (d.d_scratch2) (void)hipFree(d.d_scratch2);
  if (d.h_scratch) (void)hipHostFree(d.h_scratch);
  d.d_scratch = d.d_scratch2 = d.h_scratch = nullptr;
  d.prime_state.store(0);
  for (auto& l : d.lane) {
    for (auto& ds : l.d_scores) if (ds) (void)hipFree(ds);
    for (auto& o : l.out) {
      if (o.text) (void)hipHostFree(o.text);
      if (o.off) (void)hipHostFree(o.off);
      if (o.len) (void)hipHostFree(o.len);
      if (o.score) (void)hipHostFree(o.score);
    }
    if (l.ctx) wfagpu_amd_destroy(l.ctx);
    l = Lane{};
  }
  if (d.up) (void)hipStreamDestroy(d.up);
  if (d.down) (void)hipStreamDestroy(d.down);
  d.up = d.down = nullptr;
  d.device = -1;
}

// The copy streams (one up per device, one down per lane) are created with the highest priority the device offers: the
// runtime keeps a pool of hardware queues per priority level and maps the streams of a level onto its few queues round
// robin -- seven streams of one level had two compute lanes sharing a queue (their kernels then run one after the other), and
// which two depended on what else the process had created before (torch's streams: bench.py saw 37.5 ms where a fresh
// process saw 35.0).  With the copies in a pool of their own the lanes' streams get a queue each.
hipError_t copy_stream(hipStream_t* s) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

// The cached state of `slot` for `device`.  Streams and lanes come up through bring_up_fn / ensure_item below.
int acquire_dev(int slot, int device, int sharers, const wfagpu_amd_launch_config_t& cfg, DevState** out) {
  if (slot < 0 || slot >= MAX_DEV) return -1;
  DevState& d = g_dev[slot];
  if (d.device >= 0 && (d.device != device || d.sharers != sharers || memcmp(&d.tuning, &cfg.tuning, sizeof(d.tuning)) != 0 ||
                        d.arena_limit_cfg != cfg.arena_limit_bytes))
    release_dev(d);
  HIP_OK(hipSetDevice(device));
  if (d.device < 0) {
    d.device = device;
    d.sharers = sharers;
    d.tuning = cfg.tuning;
    d.arena_limit_cfg = cfg.arena_limit_bytes;
    if (hipMemGetInfo(&d.free_at_creation, &d.total_mem) != hipSuccess) d.free_at_creation = (size_t)16 << 30;
  }
  *out = &d;
  return 0;
}

// (creators run under create_mu and publish their item under bring_mu: a waiter never sees a half-made one, and the
// mutex the waiters need is never held across a 10-30 ms driver call)
template <typename T> void publish(DevState& d, T& slot, T value) {
  { std::lock_guard<std::mutex> l(d.bring_mu); slot = value; }
  d.bring_cv.notify_all();
}
// Lane k of a device: a context (stream, events, counters).
int make_lane(DevState& d, int k, const wfagpu_amd_launch_config_t& cfg) {
  Lane& l = d.lane[k];
  const int lanes = MAX_LANES, sharers = d.sharers;      // (every lane the slot can ever have: the caps do not depend on the call)
  wfagpu_amd_config_t c{};
  c.device = d.device;
  c.tuning = cfg.tuning;
  // The backtrace arena is kept between calls, and it is sized ONCE: by what the lane's first batch is expected to need, up to
  // the lane's share of the device -- all lanes of all slots of a device together never claim more than half of the memory that
  // was free when the slot was created; a batch that needs more runs in several passes.  (Rounds 2-4 started every arena at a
  // cap of 4 GiB that doubled after every call that had needed several passes, on the belief that fresh device memory costs
  // ~33 ms per GiB at first touch.  It does not -- 0.2 ms per GiB, profiles/r04/touch_probe.txt -- and the doubling was worse
  // than useless: the call that regrows frees its arena and allocates a bigger one, and a large hipMalloc right behind a large
  // hipFree waits for the driver to wipe what was released: a warm call of 2.48 s among calls of 77 ms, BENCH_r04 configs.cfg5.
  // hipMalloc itself is 0.3 ms whatever the size, profiles/r02/malloc_probe.txt.)
  const size_t share = d.free_at_creation / 2 / (size_t)std::max(1, sharers * std::max(lanes, MAX_LANES - 1));
  if (cfg.arena_limit_bytes) {
    c.arena_limit_bytes = cfg.arena_limit_bytes;      // fixed cap
    c.arena_limit_max_bytes = 0;
  } else {
    c.arena_limit_bytes = std::min<size_t>((size_t)48 << 30, std::max<size_t>(share, (size_t)64 << 20));
    c.arena_limit_max_bytes = 0;
  }
  wfagpu_amd_ctx_t* ctx = nullptr;
  if (wfagpu_amd_create(&ctx, &c)) return -1;
  publish(d, l.ctx, ctx);
  return 0;
}

// ---- bringing a device up ------------------------------------------------------------------------------------------------
// What a cold call pays for, measured with rocprofv3 --hip-trace on the CLI (profiles/r04/coldtrace.txt): every stream of a
// process is a hardware queue to create -- 9-35 ms each, and the creations of concurrent threads SERIALISE in the driver
// (round 3: seven streams, 190 ms of queue creation spread over the first 90 ms of a 145 ms call) --, the first copy in
// each direction sets the copy path up (~15 ms), the first launch from each code object loads it (~23 ms for the wavefront
// kernels').  So (1) one D2H stream per device instead of one per lane: five queues, not seven; (2) ONE thread creates them,
// in the order they are needed -- upload stream, lane 0, the other lanes, download stream -- while the pipeline's threads
// wait for exactly their item (ensure_item) and the first batch is being packed; (3) the first lane that is ready loads the
// code objects while it waits for the first upload; (4) the same thread is started in the BACKGROUND the first time the
// process asks this library about its devices (get_num_cuda_devices & co.: the CLI does before it reads its input,
// wfagpu_set_default_options does for the API -- the moment the reference creates its CUDA context, cudaGetDeviceCount):
// a call that comes later finds the device up.  wfagpu_amd_release_cache() (and bench.py's cold leg) undo all of it.
int create_up(DevState& d) {
  hipStream_t s = nullptr;
  HIP_OK(copy_stream(&s));
  // (the first pageable copy of a process pays ~15 ms of set-up: a small one now, not the first batch later)
  static char warm_src[1 << 16];
  if (!d.d_scratch) HIP_OK(hipMalloc(&d.d_scratch, sizeof(warm_src)));
  HIP_OK(hipMemcpyAsync(d.d_scratch, warm_src, sizeof(warm_src), hipMemcpyHostToDevice, s));
  publish(d, d.up, s);
  return 0;
}
int create_down(DevState& d) {
  hipStream_t s = nullptr;
  HIP_OK(copy_stream(&s));
  if (!d.h_scratch) HIP_OK(hipHostMalloc(&d.h_scratch, 4096, hipHostMallocDefault));
  if (!d.d_scratch2) HIP_OK(hipMalloc(&d.d_scratch2, 1 << 16));
  HIP_OK(hipMemcpyAsync(d.h_scratch, d.d_scratch2, 4096, hipMemcpyDeviceToHost, s));
  publish(d, d.down, s);
  return 0;
}
// code objects: whoever gets here first loads them (on lane k's stream), the others go on (a launch that needs a code
// object that is still loading waits inside the runtime)
void prime_once(DevState& d, int k) {
  int expect = 0;
  if (!d.lane[k].ctx || !d.prime_state.compare_exchange_strong(expect, 1)) return;
  (void)wfagpu_amd_prime(d.lane[k].ctx);
  d.prime_state.store(2);
}
// An item of the device state, created by the bring-up thread if it gets there first -- the caller then waits for exactly
// that item --, else by the caller.
template <typename Ready, typename Create> int ensure_item(DevState& d, Ready ready, Create create) {
  {
    std::unique_lock<std::mutex> l(d.bring_mu);
    d.bring_cv.wait(l, [&] { return ready() || !d.bring_active; });
    if (ready()) return 0;
  }
  std::lock_guard<std::mutex> c(d.create_mu);
  { std::lock_guard<std::mutex> l(d.bring_mu); if (ready()) return 0; }
  return create();
}
void bring_up_fn(DevState* dp, wfagpu_amd_launch_config_t cfg, int lanes, bool prime) {
  DevState& d = *dp;
  bool ok = hipSetDevice(d.device) == hipSuccess;
  auto step = [&](auto ready, auto create) {
    if (!ok) return;
    std::lock_guard<std::mutex> c(d.create_mu);
    bool have;
    { std::lock_guard<std::mutex> l(d.bring_mu); have = ready(); }
    if (!have) ok = create() == 0;
  };
  step([&] { return d.up != nullptr; }, [&] { return create_up(d); });
  d.bring_clock[0] = now_ms();
  for (int k = 0; k < lanes; ++k) { step([&] { return d.lane[k].ctx != nullptr; }, [&] { return make_lane(d, k, cfg); }); d.bring_clock[1 + k] = now_ms(); }
  step([&] { return d.down != nullptr; }, [&] { return create_down(d); });
  d.bring_clock[1 + MAX_LANES] = now_ms();
  if (ok && prime) prime_once(d, 0);      // (background bring-up: no compute lane is waiting to do it)
  { std::lock_guard<std::mutex> l(d.bring_mu); d.bring_active = false; }
  d.bring_cv.notify_all();
}
// Starts the bring-up thread of a slot unless everything `lanes` lanes need is there already.  (g_dev_mu[slot] held.)
void start_bring(DevState& d, const wfagpu_amd_launch_config_t& cfg, int lanes, bool prime) {
  // (one is still running -- the background bring-up of a process that aligns right after its first device query: its items are
  // awaited one by one by the threads that need them, whatever it does not make they make themselves: ensure_item)
  { std::lock_guard<std::mutex> l(d.bring_mu); if (d.bring_active) return; }
  join_bring(d);
  bool complete = d.up && d.down;
  for (int k = 0; k < lanes; ++k) complete = complete && d.lane[k].ctx;
  if (complete) return;
  { std::lock_guard<std::mutex> l(d.bring_mu); d.bring_active = true; }
  join_all_bring_at_exit();
  d.bring_thread = std::thread(bring_up_fn, &d, cfg, lanes, prime);
}

template <typename T> int grow_pinned(T** p, size_t want_elems) {
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  HIP_OK(hipHostMalloc(reinterpret_cast<void**>(p), want_elems * sizeof(T), hipHostMallocDefault));
  return 0;
}

struct BatchPlan {
  size_t from, to;          // pairs
  size_t lo = 0, span = 0;  // bytes of the caller's sequence buffer the batch covers: [lo, lo + span)
  size_t packed_bytes = 0;
  unsigned max_len = 0;
  bool host_packed = false; // the batch goes up as 2-bit words packed on the host (else as ASCII, packed by the kernel)
};

// One mutex + condition variable for all the per-batch stage flags of a device's pipeline.
struct Flags {
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> rc{0};
  void set(std::vector<char>& f, int i) { { std::lock_guard<std::mutex> l(mu); f[i] = 1; } cv.notify_all(); }
  // false: the pipeline has failed (the flag changes under the mutex the waiters test it under: no wake-up can be lost)
  bool wait(const std::vector<char>& f, int i) {
    if (i < 0) return rc.load() == 0;
    std::unique_lock<std::mutex> l(mu);
    cv.wait(l, [&] { return f[i] != 0 || rc.load() != 0; });
    return rc.load() == 0;
  }
  void fail(int code) { { std::lock_guard<std::mutex> l(mu); if (rc.load() == 0) rc.store(code); } cv.notify_all(); }
};

// One device's slice of a call.
int run_device(const CallArgs& a, Shard& sh) {
  if (sh.from >= sh.to) return 0;
  const double t_begin = now_ms();
  if (sh.slot < 0 || sh.slot >= MAX_DEV) return -1;
  std::lock_guard<std::mutex> guard(g_dev_mu[sh.slot]);

  // ---- batches -------------------------------------------------------------------------------------------------------
  const size_t n_all = sh.to - sh.from;
  size_t bs = a.opt.batch_size ? std::min<size_t>(a.opt.batch_size, n_all) : n_all;
  bs = std::max<size_t>(1, bs);
  // a single huge batch is cut so that the stages have something to overlap: the call ends one batch's compute + copy +
  // scatter after the last byte has crossed PCIe, so smaller batches shorten the tail -- and add per-batch overhead on
  // the device (measured, 1M x 1 kbp pairs, two lanes: 16 batches 43.8 ms, 24: 48.5; tapering the last two: 45.4)
  std::vector<BatchPlan> plan;
  auto add_batches = [&](size_t from, size_t to, size_t step) {
    for (; from < to; from += step) { BatchPlan b{}; b.from = from; b.to = std::min(to, from + step); plan.push_back(b); }
  };
  // "big" = worth a pipeline: many pairs, or many bytes (long reads: 16k x 10 kbp pairs are 330 MB)
  size_t slice_bytes = 0;
  {
    const sequence_pair_t& m0 = a.meta[sh.from];
    const sequence_pair_t& m1 = a.meta[sh.to - 1];
    const size_t lo0 = std::min(m0.pattern_offset, m0.text_offset), hi0 = std::max(m1.pattern_offset + m1.pattern_len, m1.text_offset + m1.text_len);
    slice_bytes = hi0 > lo0 ? hi0 - lo0 : 0;      // (an estimate: the records of a call are laid out in order by every known caller)
  }
  const bool big = n_all >= ((size_t)1 << 17) || (slice_bytes >= ((size_t)128 << 20) && n_all >= 256);
  // 2-bit packing on the host (a quarter of the bytes over PCIe) when this device's share of the host threads allows
  const bool host_pack = a.cfg.host_pack > 0 || (a.cfg.host_pack == 0 && sh.host_threads >= 4 && big);
  const unsigned pack_threads = a.cfg.host_pack_threads > 0 ? (unsigned)a.cfg.host_pack_threads
                                                            : std::max(2u, std::min(12u, sh.host_threads * 3u / 4u));
  const unsigned prep_threads = host_pack ? pack_threads : std::max(1u, std::min(8u, sh.host_threads / 2u));
  // (a call that is not big but not tiny either -- BASELINE configs[1]: 100k x 150 bp, 31 MB -- still gains from three or four batches:
  // record sweep, upload, kernels and scatter of ONE batch are a chain of 0.45 + 0.7 + 0.1 + 0.15 ms, cut in four the upload of a batch
  // runs under the sweep of the next: 1.46 -> 1.14 ms host to host, with CIGARs 1.87 -> 1.7 at three; more batches lose again)
  const bool mid = !big && n_all >= ((size_t)1 << 15) && slice_bytes >= ((size_t)8 << 20);
  if (bs == n_all && (big || mid || a.cfg.batches_per_device > 1)) {
    // (few long pairs: batches of >= 32 MB and >= 8192 pairs -- smaller ones tune no score budgets, csrc/wfa_host.hip, and
    // run twice as long: 16k x 10 kbp pairs, host to host: one batch 25.1 ms, two 22.0, four 43.7)
    const size_t cut = a.cfg.batches_per_device > 0 ? (size_t)a.cfg.batches_per_device
                     : mid ? (a.cigar ? 3 : 4)
                     : n_all >= ((size_t)1 << 17) ? 16 : std::max<size_t>(1, std::min<size_t>(16, std::min(slice_bytes >> 25, n_all >> 13)));
    bs = (n_all + cut - 1) / cut;
    add_batches(sh.from, sh.to, bs);
  } else {
    add_batches(sh.from, sh.to, bs);
  }
  const int nb = (int)plan.size();
  // Two lanes for big calls: the kernels of one fill the gaps (host round trips between kernel tiers, backtrace tails,
  // result copies) of the other.  Three when the sequences go up packed: the call is then bound by the kernels, not by
  // PCIe, and a third lane closes what two leave open (1M x 1 kbp pairs: 36.9 -> 35.3 ms; four lanes: the same).
  // (smaller first batches -- an earlier start for the kernels -- gain nothing: a small batch is mostly host round trips)
  int K = a.cfg.lanes_per_device > 0 ? a.cfg.lanes_per_device : ((n_all >= ((size_t)1 << 18) || (big && slice_bytes >= ((size_t)256 << 20))) ? (host_pack ? 3 : 2) : 1);
  K = std::max(1, std::min({K, MAX_LANES, nb}));

  DevState* dp = nullptr;
  if (acquire_dev(sh.slot, sh.device, sh.sharers, a.cfg, &dp)) return -1;
  DevState& d = *dp;
  HIP_OK(hipSetDevice(sh.device));
  // streams and lanes that are not there yet come up in the order they are needed, under the packing of the first batch
  start_bring(d, a.cfg, K, false);
  // (every way out of this function -- the error returns below included -- leaves no bring-up thread behind)
  struct BringGuard { DevState& d; ~BringGuard() { join_bring(d); } } bring_guard{d};
  const double t_created = now_ms();

  // ---- input slots ---------------------------------------------------------------------------------------------------
  // As many as the batches when the device has the room (then no upload ever waits for a computation), else a ring.
  int R = nb;
  {
    size_t pool = a.cfg.input_pool_bytes;
    if (!pool) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
      size_t have = 0;
      for (const auto& s : d.in) have += s.seq_cap + s.packed_cap + s.meta_cap * sizeof(sequence_pair_t);
      pool = std::min<size_t>((size_t)24 << 30, (free_b + have) / 4 / (size_t)std::max(1, sh.sharers));
    }
    // (estimate from the first batch: the pairs of a call are of similar size; the slots grow on demand anyway)
    const sequence_pair_t& m0 = a.meta[plan[0].from];
    const sequence_pair_t& m1 = a.meta[plan[0].to - 1];
    const size_t lo0 = std::min(m0.pattern_offset, m0.text_offset), hi0 = std::max(m1.pattern_offset + m1.pattern_len, m1.text_offset + m1.text_len);
    const size_t per_batch = (hi0 > lo0 ? hi0 - lo0 : 0) + (plan[0].to - plan[0].from) * sizeof(sequence_pair_t) + 4096;
    R = (int)std::max<size_t>((size_t)std::min(nb, K + 2), std::min<size_t>((size_t)nb, pool / std::max<size_t>(1, per_batch)));
  }
  if ((int)d.in.size() < R) d.in.resize(R);
  while ((int)d.up_done.size() < nb) {
    hipEvent_t e = nullptr;
    HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    d.up_done.push_back(e);
  }
  while ((int)d.down_done.size() < nb) {
    hipEvent_t e = nullptr;
    HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    d.down_done.push_back(e);
  }

  Flags fl;
  std::vector<char> prepped(nb, 0), uploaded(nb, 0), computed(nb, 0), downloaded(nb, 0), scattered(nb, 0);
  // what a computed batch left on the device (valid through the lane's next batch)
  struct BatchOut { const int32_t* d_scores = nullptr; const char* d_text = nullptr; const unsigned long long* d_off = nullptr;
                    const unsigned int* d_len = nullptr; unsigned long long text_bytes = 0; };
  std::vector<BatchOut> bout(nb);
  struct BatchClock { double prep0 = 0, prep1 = 0, up0 = 0, up1 = 0, dev0 = 0, dev1 = 0, d2h1 = 0, sc0 = 0, sc1 = 0; };      // (timing >= 2: ms since the start of the slice)
  std::vector<BatchClock> clk(nb);
  struct StageTimes { double prep = 0, pack = 0, up = 0, up_wait = 0, dev = 0, dev_wait = 0, d2h = 0, scatter = 0, check = 0; };
  double prep0_detail[3] = {0, 0, 0};      // (timing: batch 0's record sweep, staging allocation, packing)
  StageTimes t_prep_thread, t_up_thread;
  std::vector<StageTimes> t_lane(K), t_scat(K);

  // ---- stage 0: spans and packed offsets (host only) -----------------------------------------------------------------
  std::thread prepper([&] {
    if (host_pack && hipSetDevice(sh.device) != hipSuccess) { fl.fail(-1); return; }
    for (int i = 0; i < nb; ++i) {
      if (fl.rc.load()) return;
      const double t0 = now_ms();
      BatchPlan& b = plan[i];
      const size_t n = b.to - b.from;
      // Two parallel sweeps over the records (10M short reads: 30 ms on one thread -- more than their kernels take):
      // (1) per strip: span of the batch inside the caller's buffer (lib/align.cu:80-93 takes it from the first/last record;
      // scanning is robust to any record order), longest sequence, packed bytes; (2) per strip, from the prefix of those
      // sizes: the packed offsets, written into the caller's metadata like the reference (lib/align.cu:103-115,363-377),
      // relative to the batch -- wfagpu_amd_fill_packed_offsets' assignment -- and, when the batch is packed on the host,
      // the words themselves.
      const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)prep_threads, n, std::max(n >> 15, slice_bytes / std::max<size_t>(1, n_all) * n >> 22) + 1}));
      struct Strip { size_t lo = SIZE_MAX, hi = 0, bytes = 0; unsigned max_len = 0; size_t outside = SIZE_MAX; };
      std::vector<Strip> strip(nt);
      auto words = [](size_t len) { return (len + 15) / 16 + 1; };
      parallel_for(nt, [&](unsigned t) {
        Strip st;
        const size_t j1 = b.from + n * (t + 1) / nt;
        for (size_t j = b.from + n * t / nt; j < j1; ++j) {
          const sequence_pair_t& m = a.meta[j];
          // (a record that points outside [0, sequences_buffer_size) would be an out-of-bounds read on the device -- pack
          // kernel, byte-compare tiers -- and in the -c checker: the call fails below instead)
          if (st.outside == SIZE_MAX && (m.pattern_offset > a.seq_bytes || m.pattern_len > a.seq_bytes - m.pattern_offset ||
                                         m.text_offset > a.seq_bytes || m.text_len > a.seq_bytes - m.text_offset)) st.outside = j;
          st.lo = std::min(st.lo, std::min(m.pattern_offset, m.text_offset));
          st.hi = std::max(st.hi, std::max(m.pattern_offset + m.pattern_len, m.text_offset + m.text_len));
          st.max_len = std::max(st.max_len, (unsigned)std::max(m.pattern_len, m.text_len));
          st.bytes += 4 * (words(m.pattern_len) + words(m.text_len));
        }
        strip[t] = st;
      });
      if (i == 0) prep0_detail[0] = now_ms() - t0;
      size_t lo = SIZE_MAX, hi = 0, packed_bytes = 0;
      unsigned max_len = 0;
      std::vector<size_t> strip_off(nt);
      for (unsigned t = 0; t < nt; ++t) {
        lo = std::min(lo, strip[t].lo); hi = std::max(hi, strip[t].hi); max_len = std::max(max_len, strip[t].max_len);
        strip_off[t] = packed_bytes; packed_bytes += strip[t].bytes;
        if (strip[t].outside != SIZE_MAX) {
          LOG_ERROR("Sequence record %zu points outside the sequence buffer (%zu bytes).", strip[t].outside, a.seq_bytes);
          fl.fail(-1);
          return;
        }
      }
      lo &= ~(size_t)3;
      hi = std::min(a.seq_bytes, (hi + 4) & ~(size_t)3);
      b.lo = lo; b.span = hi > lo ? hi - lo : 0; b.max_len = max_len;
      b.packed_bytes = packed_bytes;
      uint32_t* stage = nullptr;
      double tp0 = 0;
      // A/B (wfagpu_amd_launch_config_t::ascii_every >= 2): every n-th batch of a big call goes up as ASCII -- four times the bytes, no
      // packing -- and the wavefront kernels pack it while they stage it.  The idea: packing all sixteen batches of 1M x 1 kbp pairs takes
      // twelve threads 24-27 ms, as long as the kernels take.  Measured (scratch/ab_ascii_every.sh, alternating processes, warm calls):
      // all packed 34.3-35.3 ms, every 3rd / 2nd / 4th batch as ASCII 35.6-36.5 / 35.8-36.4 / 35.7-36.4: the packing does not set the
      // pace (profiles/r06/ab_ascii_every.txt).  Off by default.
      const int ascii_every = a.cfg.ascii_every >= 2 ? a.cfg.ascii_every : 0;
      const bool pack_this = host_pack && !(a.cfg.host_pack == 0 && big && nb >= 6 && ascii_every && b.max_len >= 512u && (i % ascii_every) == 0);
      if (pack_this) {
        tp0 = now_ms();
        // (the staging buffer last carried batch i - STAGE_RING: its copy must have left the host)
        if (i >= STAGE_RING) {
          if (!fl.wait(uploaded, i - STAGE_RING)) return;
          if (hipEventSynchronize(d.up_done[i - STAGE_RING]) != hipSuccess) { fl.fail(-1); return; }
        }
        HostStage& hs = d.stage[i % STAGE_RING];
        if (b.packed_bytes + 64 > hs.cap) {
          free(hs.p);
          hs.cap = b.packed_bytes + b.packed_bytes / 8 + 64;
          hs.p = static_cast<uint32_t*>(malloc(hs.cap));
          if (!hs.p) { hs.cap = 0; LOG_ERROR("Can not allocate the packing buffer"); fl.fail(-1); return; }
        }
        stage = hs.p;
        if (i == 0) prep0_detail[1] = now_ms() - tp0;
      }
      std::atomic<int> bad{0};
      const double t_pack0 = now_ms();
      parallel_for(nt, [&](unsigned t) {
        const size_t j0 = b.from + n * t / nt, j1 = b.from + n * (t + 1) / nt;
        if (wfagpu_host_pack_strip(a.seq, a.seq_bytes, a.meta + j0, j1 - j0, strip_off[t], stage)) bad.store(1);
      });
      if (i == 0) prep0_detail[2] = now_ms() - t_pack0;
      if (pack_this) {
        // (a byte outside ACGT: those pairs need their ASCII on the device, the whole batch goes up as it is)
        b.host_packed = bad.load() == 0;
        t_prep_thread.pack += now_ms() - tp0;
      }
      t_prep_thread.prep += now_ms() - t0;
      clk[i].prep0 = t0 - t_begin; clk[i].prep1 = now_ms() - t_begin;
      fl.set(prepped, i);
    }
  });

  // ---- stage 1: H2D, one batch after the other -------------------------------------------------------------------------
  std::thread uploader([&] {
    if (hipSetDevice(sh.device) != hipSuccess) { fl.fail(-1); return; }
    auto ok = [&](hipError_t e, const char* what) { if (e != hipSuccess) { LOG_ERROR("HIP call %s failed: %s", what, hipGetErrorString(e)); fl.fail(-1); return false; } return true; };
    if (ensure_item(d, [&] { return d.up != nullptr; }, [&] { return create_up(d); })) { fl.fail(-1); return; }
    for (int i = 0; i < nb; ++i) {
      double t0 = now_ms();
      if (!fl.wait(prepped, i)) return;
      if (!fl.wait(computed, i - R)) return;          // (a ring of slots: the one of batch i - R must have been consumed)
      t_up_thread.up_wait += now_ms() - t0; t0 = now_ms();
      clk[i].up0 = t0 - t_begin;
      const BatchPlan& b = plan[i];
      const size_t n = b.to - b.from;
      InSlot& in = d.in[i % R];
      if (b.host_packed) {
        if (b.packed_bytes + 64 > in.packed_cap) {
          if (in.d_packed) (void)hipFree(in.d_packed);
          in.d_packed = nullptr; in.packed_cap = b.packed_bytes + b.packed_bytes / 8 + 64;
          if (!ok(hipMalloc(&in.d_packed, in.packed_cap), "hipMalloc(packed sequences)")) return;
        }
      } else if (b.span + 16 > in.seq_cap) {
        if (in.d_seq) (void)hipFree(in.d_seq);
        in.d_seq = nullptr; in.seq_cap = b.span + b.span / 8 + 16;
        if (!ok(hipMalloc(&in.d_seq, in.seq_cap), "hipMalloc(sequences)")) return;
      }
      if (n > in.meta_cap) {
        if (in.d_meta) (void)hipFree(in.d_meta);
        in.d_meta = nullptr; in.meta_cap = n + n / 8;
        if (!ok(hipMalloc(&in.d_meta, in.meta_cap * sizeof(sequence_pair_t)), "hipMalloc(metadata)")) return;
      }
      // (the records go up as they are: the kernels see the sequences through a base pointer moved back by b.lo, so the
      // caller's offsets need no rebasing and no host copy)
      if (b.host_packed) {
        if (b.packed_bytes && !ok(hipMemcpyAsync(in.d_packed, d.stage[i % STAGE_RING].p, b.packed_bytes, hipMemcpyHostToDevice, d.up), "H2D packed sequences")) return;
      } else if (b.span && !ok(hipMemcpyAsync(in.d_seq, a.seq + b.lo, b.span, hipMemcpyHostToDevice, d.up), "H2D sequences")) return;
      if (!ok(hipMemcpyAsync(in.d_meta, a.meta + b.from, n * sizeof(sequence_pair_t), hipMemcpyHostToDevice, d.up), "H2D metadata")) return;
      // (no synchronisation here: the copies of the next batch follow back to back; the lane that takes this batch
      // waits for the event)
      if (!ok(hipEventRecord(d.up_done[i], d.up), "H2D event")) return;
      t_up_thread.up += now_ms() - t0;
      fl.set(uploaded, i);
    }
  });

  // ---- stage 3: staging -> the caller's records (+ -c), one thread per lane --------------------------------------------
  const unsigned lane_threads = std::max(1u, sh.host_threads / (unsigned)K);
  // The lanes TAKE batches (the next one that has not been taken) instead of owning every K-th: the kernels of
  // concurrent lanes do not share the device evenly (1M x 1 kbp pairs, three lanes: one lane went through its batches in
  // 2.6 ms each while the others took 8 and was idle for the last third of the call).
  std::atomic<int> next_batch{0};
  std::vector<std::vector<int>> lane_batches(K);      // what lane k has computed, in order (under fl.mu)
  std::vector<char> lane_done(K, 0);
  auto scatter_lane = [&](int k) {
    // (this thread issues the D2H copies of its lane: every thread that talks to HIP selects the slice's device first)
    if (hipSetDevice(sh.device) != hipSuccess) { fl.fail(-1); return; }
    if (ensure_item(d, [&] { return d.down != nullptr; }, [&] { return create_down(d); })) { fl.fail(-1); return; }
    {
      // pinned staging for the fixed-size results of both output sets, before the first batch is through (pinned
      // memory costs ~0.2 ms per MiB: not on the first results' way home)
      const size_t n0 = plan[0].to - plan[0].from, cap = n0 + n0 / 8;
      for (auto& o : d.lane[k].out) {
        if (n0 > o.n_cap) { if (grow_pinned(&o.score, cap)) { fl.fail(-1); return; } o.n_cap = cap; }
        if (a.cigar && n0 > o.cig_cap) { if (grow_pinned(&o.off, cap) || grow_pinned(&o.len, cap)) { fl.fail(-1); return; } o.cig_cap = cap; }
      }
    }
    for (int j = 0;; ++j) {
      int i = -1;
      {
        std::unique_lock<std::mutex> l(fl.mu);
        fl.cv.wait(l, [&] { return (int)lane_batches[k].size() > j || lane_done[k] != 0 || fl.rc.load() != 0; });
        if (fl.rc.load() != 0) return;
        if ((int)lane_batches[k].size() <= j) break;
        i = lane_batches[k][j];
      }
      double t0 = now_ms();
      const BatchPlan& b = plan[i];
      const size_t n = b.to - b.from;
      Lane& L = d.lane[k];
      Lane::Out& o = L.out[j & 1];
      {
        // results of the batch -> pinned staging, on the lane's copy stream: the lane's context is already running its next batch
        auto okh = [&](hipError_t e, const char* what) { if (e != hipSuccess) { LOG_ERROR("HIP call %s failed: %s", what, hipGetErrorString(e)); fl.fail(-1); return false; } return true; };
        const BatchOut& bo = bout[i];
        if (n > o.n_cap) {
          const size_t cap = n + n / 8;
          if (grow_pinned(&o.score, cap)) { fl.fail(-1); return; }
          o.n_cap = cap;
        }
        if (a.cigar && n > o.cig_cap) {
          const size_t cap = n + n / 8;
          if (grow_pinned(&o.off, cap) || grow_pinned(&o.len, cap)) { fl.fail(-1); return; }
          o.cig_cap = cap;
        }
        if (!okh(hipMemcpyAsync(o.score, bo.d_scores, n * sizeof(int32_t), hipMemcpyDeviceToHost, d.down), "D2H scores")) return;
        if (a.cigar) {
          if (bo.text_bytes + 1 > o.text_cap) {
            const size_t cap = (size_t)bo.text_bytes + (size_t)bo.text_bytes / 8 + 4096;
            if (grow_pinned(&o.text, cap)) { fl.fail(-1); return; }
            o.text_cap = cap;
          }
          if (!okh(hipMemcpyAsync(o.off, bo.d_off, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, d.down), "D2H offsets")) return;
          if (!okh(hipMemcpyAsync(o.len, bo.d_len, n * sizeof(unsigned int), hipMemcpyDeviceToHost, d.down), "D2H lengths")) return;
          if (bo.text_bytes && !okh(hipMemcpyAsync(o.text, bo.d_text, bo.text_bytes, hipMemcpyDeviceToHost, d.down), "D2H text")) return;
        }
        // (the device's one download stream: this batch's copies -- and whatever other lanes queued before them -- have landed)
        if (!okh(hipEventRecord(d.down_done[i], d.down), "D2H event")) return;
        if (!okh(hipEventSynchronize(d.down_done[i]), "D2H sync")) return;
        t_scat[k].d2h += now_ms() - t0;
        clk[i].d2h1 = now_ms() - t_begin;
        fl.set(downloaded, i);
        t0 = now_ms();
      }
      clk[i].sc0 = t0 - t_begin;
      std::atomic<int> bad{0};
      // Every cigar.buffer stays the caller's own, individually free()-able allocation (the ABI: lib/alignment_results.c).
      // One that is too small grows to the longest text of the batch, so that the following calls of a process (same
      // reads, similar CIGARs) find every buffer big enough: no realloc in the steady state.
      unsigned max_len = 0;
      if (a.cigar) for (size_t q = 0; q < n; ++q) if (o.len[q] != 0xFFFFFFFFu) max_len = std::max(max_len, o.len[q]);
      const size_t grow_to = (size_t)max_len + 1;
      auto work = [&](size_t j0, size_t j1) {
        for (size_t q = j0; q < j1; ++q) {
          wfa_alignment_result_t& r = a.results[b.from + q];
          r.error = (unsigned int)o.score[q];
          if (!a.cigar) continue;
          if (o.len[q] == 0xFFFFFFFFu) { LOG_ERROR("CIGAR recovery failed for pair %zu", b.from + q); bad.store(1); continue; }
          wfa_cigar_t& cg = r.cigar;
          const size_t need = (size_t)o.len[q] + 1;
          if (need > cg.buffer_size || !cg.buffer) {
            char* nbuf = static_cast<char*>(realloc(cg.buffer, grow_to));
            if (!nbuf) { LOG_ERROR("Can not realloc CIGAR buffer"); exit(-1); }   // utils/wfa_cpu.c:77-80
            cg.buffer = nbuf; cg.buffer_size = grow_to;
          }
          memcpy(cg.buffer, o.text + o.off[q], need);
          cg.last_free_position = o.len[q];
        }
      };
      // strips of consecutive pairs per thread
      const unsigned nt = (unsigned)std::min<size_t>(std::min(8u, lane_threads), a.cigar ? (n + 8191) / 8192 : (n + 131071) / 131072);
      parallel_for(std::max(1u, nt), [&](unsigned t) { work(n * t / std::max(1u, nt), n * (t + 1) / std::max(1u, nt)); });
      t_scat[k].scatter += now_ms() - t0;
      if (bad.load()) { fl.fail(-1); return; }
      if (a.check) { const double t1 = now_ms(); check_batch(a, b.from, b.to, i, lane_threads); t_scat[k].check += now_ms() - t1; }
      clk[i].sc1 = now_ms() - t_begin;
      fl.set(scattered, i);
    }
  };

  // ---- stage 2: the kernels + D2H, K lanes on alternate batches ----------------------------------------------------------
  auto compute_lane = [&](int k) -> int {
    HIP_OK(hipSetDevice(sh.device));
    if (ensure_item(d, [&] { return d.lane[k].ctx != nullptr; }, [&] { return make_lane(d, k, a.cfg); })) return -1;
    prime_once(d, k);      // (code objects: the first lane that is up loads them while the first batch is on its way)
    Lane& L = d.lane[k];
    std::vector<int> mine;
    for (int j = 0;; ++j) {
      const int i = next_batch.fetch_add(1);
      if (i >= nb) break;
      mine.push_back(i);
      double t0 = now_ms();
      if (!fl.wait(uploaded, i)) return fl.rc.load();
      HIP_OK(hipEventSynchronize(d.up_done[i]));
      t_lane[k].dev_wait += now_ms() - t0; t0 = now_ms();
      clk[i].up1 = clk[i].dev0 = t0 - t_begin;
      const BatchPlan& b = plan[i];
      const size_t n = b.to - b.from;
      const InSlot& in = d.in[i % R];
      int32_t*& d_sc = L.d_scores[j & 1];
      // (the output set this batch writes -- score buffer included -- was last used by the lane's batch before last: its
      // download must be over before the buffer is written or, when it has to grow, freed)
      if (j >= 2 && !fl.wait(downloaded, mine[j - 2])) return fl.rc.load();
      if (n > L.scores_cap[j & 1]) {
        if (d_sc) (void)hipFree(d_sc);
        d_sc = nullptr; L.scores_cap[j & 1] = n + n / 8;
        HIP_OK(hipMalloc(&d_sc, L.scores_cap[j & 1] * sizeof(int32_t)));
      }
      wfagpu_amd_batch_t wb{};
      if (b.host_packed) wb.d_packed = in.d_packed;
      else wb.d_sequences = reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(in.d_seq) - b.lo);     // [offset of the caller's buffer]
      wb.sequences_bytes = b.lo + b.span; wb.d_metadata = in.d_meta; wb.num_pairs = n;
      wb.packed_bytes = b.packed_bytes; wb.max_seq_len = b.max_len;
      BatchOut& bo = bout[i];
      // The batches of a call come from one stream of reads -- and so, as far as the budgets go, do the calls of a process:
      // the budgets a lane learnt from a sample stay on trial (same penalties, same max_error, same length bucket; results are
      // exact either way, and a batch in which more than 5 % of the pairs miss them makes the next one sample again).
      wfagpu_amd_hint_same_stream(L.ctx, 1);
      const int arc = wfagpu_amd_align_device(L.ctx, &wb, a.opt.penalties, a.opt.max_error, a.opt.band, a.opt.threads_per_block, a.cigar,
                                              d_sc, &bo.d_text, &bo.d_off, &bo.d_len);
      if (arc) return arc;
      bo.d_scores = d_sc;
      if (a.cigar) { wfagpu_amd_stats_t stt; wfagpu_amd_last_stats(L.ctx, &stt); bo.text_bytes = stt.text_bytes; }
      t_lane[k].dev += now_ms() - t0;
      clk[i].dev1 = now_ms() - t_begin;
      { std::lock_guard<std::mutex> l(fl.mu); computed[i] = 1; lane_batches[k].push_back(i); }
      fl.cv.notify_all();
    }
    return 0;
  };
  auto run_lane = [&](int k) {
    const int c = compute_lane(k);
    if (c) fl.fail(c);
    { std::lock_guard<std::mutex> l(fl.mu); lane_done[k] = 1; }
    fl.cv.notify_all();
  };

  std::vector<std::thread> scat, comp;
  for (int k = 0; k < K; ++k) scat.emplace_back(scatter_lane, k);
  for (int k = 1; k < K; ++k) comp.emplace_back(run_lane, k);
  run_lane(0);
  for (auto& t : comp) t.join();
  prepper.join();
  uploader.join();
  for (auto& t : scat) t.join();
  join_bring(d);
  // (the device is idle: what the lanes' contexts outgrew during the call can go without stalling anybody)
  for (int k = 0; k < K; ++k) if (d.lane[k].ctx) wfagpu_amd_trim(d.lane[k].ctx);

  wfagpu_amd_launch_stats_t& st = sh.st;
  st.total_ms = now_ms() - t_begin;
  st.acquire_ms = t_created - t_begin;
  st.prep_ms = t_prep_thread.prep; st.upload_ms = t_up_thread.up; st.upload_wait_ms = t_up_thread.up_wait;
  for (int k = 0; k < K; ++k) {
    st.device_ms += t_lane[k].dev; st.device_wait_ms += t_lane[k].dev_wait; st.d2h_ms += t_scat[k].d2h;
    st.scatter_ms += t_scat[k].scatter; st.check_ms += t_scat[k].check;
  }
  st.lanes = K; st.batches = nb;
  st.host_pack_ms = t_prep_thread.pack; st.host_pack_threads = host_pack ? (int)pack_threads : 0;
  for (const auto& b : plan) st.host_packed_batches += b.host_packed ? 1 : 0;
  if (a.cfg.timing)
    fprintf(stderr, "[wfagpu timing] device %d (slot %d): %d batches, %d lanes, %d input slots; acquire %.1f ms, prep %.1f, upload %.1f (+%.1f waiting), "
            "device %.1f (+%.1f waiting), d2h %.1f, scatter %.1f, check %.1f; total %.1f\n",
            sh.device, sh.slot, nb, K, R, st.acquire_ms, st.prep_ms, st.upload_ms, st.upload_wait_ms, st.device_ms, st.device_wait_ms,
            st.d2h_ms, st.scatter_ms, st.check_ms, st.total_ms);
  if (a.cfg.timing >= 2) {
    if (d.bring_clock[0] > t_begin) {
      fprintf(stderr, "[wfagpu timing]   bring-up: upload stream %.2f", d.bring_clock[0] - t_begin);
      for (int k = 0; k < K; ++k) fprintf(stderr, ", lane %d %.2f", k, d.bring_clock[1 + k] - t_begin);
      fprintf(stderr, ", download stream %.2f\n", d.bring_clock[1 + MAX_LANES] - t_begin);
    }
    fprintf(stderr, "[wfagpu timing]   batch 0 prep: record sweep %.2f ms, staging buffer %.2f, offsets + packing %.2f (%u threads)\n",
            prep0_detail[0], prep0_detail[1], prep0_detail[2], prep_threads);
    for (int i = 0; i < nb; ++i)
      fprintf(stderr, "[wfagpu timing]   batch %2d (%zu pairs): prep %.2f - %.2f | upload issued %.2f, landed <= %.2f | device %.2f - %.2f | d2h done %.2f | scatter %.2f - %.2f\n",
              i, plan[i].to - plan[i].from, clk[i].prep0, clk[i].prep1, clk[i].up0, clk[i].up1, clk[i].dev0, clk[i].dev1, clk[i].d2h1, clk[i].sc0, clk[i].sc1);
  }
  return fl.rc.load();
}

void launch_impl(char* seq, size_t seq_bytes, sequence_pair_t* meta, wfa_alignment_result_t* results,
                 wfa_alignment_options_t opt, bool check, bool cigar) {
  if (!seq || !meta || !results) { LOG_ERROR("Invalid buffers."); return; }
  const size_t n = opt.num_alignments;
  if (n == 0) return;
  const double t_begin = now_ms();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    LOG_ERROR("No HIP device available.");
    exit(-1);   // utils/cuda_utils.cuh:28-35: device errors are fatal
  }
  CallArgs a{seq, seq_bytes, meta, results, opt, check, cigar, {}};
  { std::lock_guard<std::mutex> l(g_cfg_mu); a.cfg = g_cfg; }
  if (a.cfg.num_devices > 0) ndev = std::min(ndev, a.cfg.num_devices);
  // virtual devices (tests): that many shards with their own threads, contexts and streams, mapped round-robin
  // onto the physical devices -- exercises the multi-device path on a single GPU
  const int physical = ndev;
  if (a.cfg.virtual_devices > 0) ndev = std::min(MAX_DEV, a.cfg.virtual_devices);
  ndev = (int)std::min<size_t>((size_t)std::min(ndev, MAX_DEV), n);
  g_check_failures.store(0);
  const unsigned host_threads = usable_host_threads();
  std::vector<Shard> shards(ndev);
  // Contiguous slices of equal WORK, not of equal count: the wavefront work of a pair grows with P x T (score^2 at a
  // given error rate), so a ragged batch cut by count would leave devices idle.  Partial sums per strip in parallel,
  // then one scan over the strip that holds each cut.
  std::vector<size_t> cut(ndev + 1, n);
  cut[0] = 0;
  if (ndev > 1) {
    auto work_of = [&](size_t i) { return (double)meta[i].pattern_len * (double)meta[i].text_len + 1024.0; };   // (+ a per-pair constant)
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(16u, host_threads), n / 65536));
    std::vector<double> part(nt, 0.0);
    parallel_for(nt, [&](unsigned t) { double s = 0; for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) s += work_of(i); part[t] = s; });
    double total = 0;
    for (double p : part) total += p;
    double acc = 0;
    int dcut = 1;
    for (unsigned t = 0; t < nt && dcut < ndev; ++t) {
      if (acc + part[t] < total * dcut / ndev) { acc += part[t]; continue; }
      for (size_t i = n * t / nt; i < n * (t + 1) / nt && dcut < ndev; ++i) {
        acc += work_of(i);
        while (dcut < ndev && acc >= total * dcut / ndev) cut[dcut++] = i + 1;
      }
    }
  }
  for (int d = 0; d < ndev; ++d) {
    shards[d].device = d % physical;
    shards[d].slot = d;
    shards[d].sharers = (ndev - 1 - (d % physical)) / physical + 1;      // slots mapped onto the same physical device
    shards[d].from = cut[d];
    shards[d].to = cut[d + 1];
    shards[d].host_threads = std::max(1u, std::min(64u, host_threads / (unsigned)ndev));
  }
  const double t_planned = now_ms();
  if (ndev == 1) {
    shards[0].rc = run_device(a, shards[0]);
  } else {
    std::vector<std::thread> th;
    // (own threads, one per device: each may be moved to the cores next to its GPU; the caller's thread is never touched)
    const bool pin = a.cfg.numa_pin > 0 || (a.cfg.numa_pin == 0 && physical > 1);
    for (int d = 0; d < ndev; ++d)
      th.emplace_back([&a, &shards, d, pin] {
        if (pin) pin_to_device_node(shards[d].device);
        shards[d].rc = run_device(a, shards[d]);
      });
    for (auto& t : th) t.join();
  }
  for (auto& s : shards)
    if (s.rc) { LOG_ERROR("Alignment failed on device %d (code %d).", s.device, s.rc); exit(-1); }
  // stage times of the busiest device
  wfagpu_amd_launch_stats_t st{};
  for (auto& s : shards) if (s.st.total_ms >= st.total_ms) st = s.st;
  st.plan_ms = t_planned - t_begin;
  st.devices = ndev;
  st.host_threads = host_threads;
  st.total_ms = now_ms() - t_begin;
  {
    std::lock_guard<std::mutex> l(g_cfg_mu);
    g_last_stats = st;
    g_last_shard_stats.clear();
    for (auto& s : shards) { s.st.devices = s.device; s.st.host_threads = s.host_threads; g_last_shard_stats.push_back(s.st); }
  }
}

}  // namespace

extern "C" {

void wfagpu_amd_configure_launch(const wfagpu_amd_launch_config_t* cfg) {
  std::lock_guard<std::mutex> l(g_cfg_mu);
  g_cfg = cfg ? *cfg : wfagpu_amd_launch_config_t{};
}

void wfagpu_amd_last_launch_stats(wfagpu_amd_launch_stats_t* out) {
  if (!out) return;
  std::lock_guard<std::mutex> l(g_cfg_mu);
  *out = g_last_stats;
}

int wfagpu_amd_last_launch_stats_device(int shard, wfagpu_amd_launch_stats_t* out) {
  std::lock_guard<std::mutex> l(g_cfg_mu);
  if (!out || shard < 0 || shard >= (int)g_last_shard_stats.size()) return -1;
  *out = g_last_shard_stats[shard];
  return 0;
}

void wfagpu_amd_set_num_devices(int n) {
  std::lock_guard<std::mutex> l(g_cfg_mu);
  g_cfg.num_devices = n < 0 ? 0 : n;
}

long wfagpu_amd_check_failures(void) { return g_check_failures.load(); }

void wfagpu_amd_release_cache(void) {
  for (int i = 0; i < MAX_DEV; ++i) {
    std::lock_guard<std::mutex> guard(g_dev_mu[i]);
    release_dev(g_dev[i]);
  }
}

void launch_alignments(char* sequences_buffer, const size_t sequences_buffer_size,
                       sequence_pair_t* const sequences_metadata,
                       wfa_alignment_result_t* const alignment_results,
                       wfa_alignment_options_t options, bool check_correctness) {
  launch_impl(sequences_buffer, sequences_buffer_size, sequences_metadata, alignment_results, options,
              check_correctness, true);
}

void launch_alignments_distance(char* sequences_buffer, const size_t sequences_buffer_size,
                                sequence_pair_t* const sequences_metadata,
                                wfa_alignment_result_t* const alignment_results,
                                wfa_alignment_options_t options, bool check_correctness) {
  launch_impl(sequences_buffer, sequences_buffer_size, sequences_metadata, alignment_results, options,
              check_correctness, false);
}

// Brings devices [first, last] up in the background.  The caller's current HIP device is left as it was found.
static void warm_devices(int first, int last) {
  wfagpu_amd_launch_config_t cfg;
  { std::lock_guard<std::mutex> l(g_cfg_mu); cfg = g_cfg; }
  if (cfg.bring_up < 0 || cfg.virtual_devices > 0) return;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return;
  if (cfg.num_devices > 0) ndev = std::min(ndev, cfg.num_devices);
  int caller_dev = -1;
  if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
  for (int dev = std::max(first, 0); dev <= std::min(last, std::min(ndev, MAX_DEV) - 1); ++dev) {
    std::lock_guard<std::mutex> guard(g_dev_mu[dev]);
    DevState* d = nullptr;
    if (acquire_dev(dev, dev, 1, cfg, &d)) break;
    start_bring(*d, cfg, MAX_LANES - 1, true);
  }
  if (caller_dev >= 0) (void)hipSetDevice(caller_dev);      // (acquire_dev selects the device it sets up)
}

// Explicit: every device a call would be sharded over (wfagpu_amd_launch_config_t::num_devices; the CLI calls this).
void wfagpu_amd_warmup(void) { warm_devices(0, MAX_DEV - 1); }

// (a slot's lock is what a call holds while it runs on the slot: run_device)
void wfagpu_amd_warmup_wait(void) {
  for (int i = 0; i < MAX_DEV; ++i) { std::lock_guard<std::mutex> guard(g_dev_mu[i]); join_bring(g_dev[i]); }
}

// The first time the process asks about its devices (the CLI: tools/aligner.c:189-204 of the reference, before it reads its
// input; the API: wfagpu_set_default_options -> get_cuda_SM_count) -- the moment the reference creates its CUDA context on
// ITS device (device 0, lib/alignment_parameters.h:77) -- the caller's CURRENT device starts coming up in the background:
// one device, not every visible one (a rank of a one-process-per-GPU job that has selected its GPU opens no context, no
// hardware queue and no VRAM on the other seven), and the caller's current device stays what it was.
static void first_device_query() {
  static std::once_flag once;
  std::call_once(once, [] {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    warm_devices(cur, cur);
  });
}

void get_num_cuda_devices(int* n) {
  if (!n) return;
  if (hipGetDeviceCount(n) != hipSuccess) *n = 0;
  if (*n > 0) first_device_query();
}

char* get_cuda_dev_name(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return strdup("unknown");
  return strdup(prop.name);
}

int get_cuda_SM_count(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  first_device_query();
  return prop.multiProcessorCount;
}

void get_cuda_capability(int dev, int* major, int* minor) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { if (major) *major = 0; if (minor) *minor = 0; return; }
  if (major) *major = prop.major;
  if (minor) *minor = prop.minor;
}

}  // extern "C"
