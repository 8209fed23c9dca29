// The C-ABI seam: launch_alignments / launch_alignments_distance
// (lib/align.cuh:35-47 of the reference, implemented there by
// lib/align.cu:42-881) and the device-query shims
// (utils/device_query.cu:27-54).
//
// Host buffers in, host results out, blocking, results in input order.  The
// pairs of a call are sharded in contiguous slices over the visible GPUs
// (one host thread + one context + one stream per device, no collective:
// pairs are independent); inside a device the slice is cut into batches of
// options.batch_size like the reference does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include "../../include/wfa_gpu_device.h"
#include "../utils/logger.h"
#include "../utils/verification.h"

namespace {

std::atomic<int> g_num_devices{0};   // 0: all visible
std::atomic<long> g_check_failures{0};   // pairs that failed the -c check in the last launch_alignments* call

// Host cores this process may use (affinity mask), the budget the slices of a call share.
unsigned usable_host_threads() {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return (unsigned)n; }
  return std::max(1u, std::thread::hardware_concurrency());
}

// Best effort: run a slice's host threads (this one and the ones it starts: they inherit the mask) on the cores of the
// NUMA node its GPU hangs off, so that the staging copies and the scatter of eight slices do not all cross the socket
// interconnect.  Intersected with the mask the process already has; any failure leaves things as
// they are.  WFAGPU_NO_NUMA_PIN=1 disables it.
void pin_to_device_node(int device) {
  if (getenv("WFAGPU_NO_NUMA_PIN")) return;
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
  int slot;          // index of the cached per-device state (== device unless WFAGPU_VIRTUAL_DEVICES is set)
  size_t from, to;   // [from, to)
  unsigned host_threads = 1;   // this slice's share of the host cores (scatter and -c workers): SURVEY.md section 8(e)
  int rc = 0;
};

struct CallArgs {
  char* seq;
  size_t seq_bytes;
  sequence_pair_t* meta;
  wfa_alignment_result_t* results;
  wfa_alignment_options_t opt;
  bool check;
  bool cigar;
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
  auto worker = [&] {
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
        const int cpu = verification_cpu_score(pattern, text, m.pattern_len, m.text_len, a.opt.penalties.x,
                                               a.opt.penalties.o, a.opt.penalties.e);
        if (cpu != dist) { LOG_ERROR("Incorrect distance (%ld). GPU=%d, CPU=%d", i, dist, cpu); ok = false; }
        sum += dist;
        if (ok) ++correct; else ++incorrect;
      }
    }
  };
  // (each slice has its share of the host cores: with 8 devices x 2 slices a pool of hardware_concurrency() threads per
  // batch and slice would oversubscribe the host 16-fold and starve the upload and scatter threads)
  unsigned nt = std::max(1u, max_threads);
  nt = (unsigned)std::min<size_t>(nt, (to - from + 15) / 16);
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  fprintf(stderr, "(Batch %d) correct=%ld Incorrect=%ld Average score=%f\n", batch_idx, correct.load(), incorrect.load(),
          (to > from) ? (double)sum.load() / (double)(to - from) : 0.0);
  g_check_failures += incorrect.load();
  return incorrect.load() ? 1 : 0;
}

// Per-device state that outlives a call: the context (stream, arena, scratch), two input buffers,
// two sets of pinned result staging.  Allocating them is most of the cost of a cold call (device memory
// ~33 ms per GiB, pinned host memory similar), so they are kept for the next call of the process and
// freed by wfagpu_amd_release_cache() or at exit by the OS.  Calls are single-threaded by contract
// (lib/aligner.h of the reference is not re-entrant); a mutex guards the cache anyway.
struct DevState {
  int device = -1;
  wfagpu_amd_ctx_t* ctx = nullptr;
  hipStream_t up = nullptr, down = nullptr;   // copy streams (H2D of the next batch, D2H of the last)
  struct In { char* d_seq = nullptr; size_t seq_cap = 0; sequence_pair_t* d_meta = nullptr; size_t meta_cap = 0; } in[2];
  int32_t* d_scores = nullptr; size_t scores_cap = 0;
  struct Out {
    char* text = nullptr; size_t text_cap = 0;
    unsigned long long* off = nullptr; unsigned int* len = nullptr; size_t cig_cap = 0;   // CIGAR calls only
    int32_t* score = nullptr; size_t n_cap = 0;
  } out[2];
};
constexpr int MAX_DEV = 64;
DevState g_dev[MAX_DEV];
std::mutex g_dev_mu[MAX_DEV];   // one per device: the shards of a call run concurrently

void release_dev(DevState& d) {
  if (d.device < 0) return;
  (void)hipSetDevice(d.device);
  for (auto& in : d.in) { if (in.d_seq) (void)hipFree(in.d_seq); if (in.d_meta) (void)hipFree(in.d_meta); in = {}; }
  if (d.d_scores) (void)hipFree(d.d_scores);
  for (auto& o : d.out) {
    if (o.text) (void)hipHostFree(o.text);
    if (o.off) (void)hipHostFree(o.off);
    if (o.len) (void)hipHostFree(o.len);
    if (o.score) (void)hipHostFree(o.score);
    o = {};
  }
  if (d.up) (void)hipStreamDestroy(d.up);
  if (d.down) (void)hipStreamDestroy(d.down);
  if (d.ctx) wfagpu_amd_destroy(d.ctx);
  d.device = -1; d.ctx = nullptr; d.up = d.down = nullptr; d.d_scores = nullptr; d.scores_cap = 0;
}

int acquire_dev(int slot, int device, DevState** out) {
  if (slot < 0 || slot >= MAX_DEV) return -1;
  DevState& d = g_dev[slot];
  if (d.device == device && d.ctx) { *out = &d; return 0; }
  HIP_OK(hipSetDevice(device));
  wfagpu_amd_config_t cfg{};
  cfg.device = device;
  // The backtrace arena is kept between calls.  Its cap starts at 4 GiB -- fresh device memory costs ~33 ms per GiB at
  // first touch, and a batch that needs more simply runs in several passes (1M x 1 kbp pairs: 18 GB of origin bytes,
  // 5 passes, 130 ms cold instead of 1.2 s) -- and doubles after every call that needed several passes, up to a quarter
  // of the free device memory or 32 GiB: a long-lived process ends up with one pass per call.
  cfg.arena_limit_bytes = (size_t)4 << 30;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)16 << 30;
    cfg.arena_limit_max_bytes = std::max<size_t>(cfg.arena_limit_bytes, std::min<size_t>((size_t)32 << 30, free_b / 4));
  }
  if (const char* e = getenv("WFAGPU_ARENA_LIMIT_MB")) { cfg.arena_limit_bytes = (size_t)atol(e) << 20; cfg.arena_limit_max_bytes = 0; }
  if (wfagpu_amd_create(&d.ctx, &cfg)) return -1;
  d.device = device;
  HIP_OK(hipStreamCreateWithFlags(&d.up, hipStreamNonBlocking));
  HIP_OK(hipStreamCreateWithFlags(&d.down, hipStreamNonBlocking));
  *out = &d;
  return 0;
}

template <typename T> int grow_pinned(T** p, size_t want_elems) {
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  HIP_OK(hipHostMalloc(reinterpret_cast<void**>(p), want_elems * sizeof(T), hipHostMallocDefault));
  return 0;
}

struct BatchPlan {
  size_t from, to, lo, span, packed_bytes;
  unsigned max_len;
  unsigned long long text_bytes = 0;
};

// One device's slice of a call as a three-stage pipeline over its batches (the reference overlaps the
// same phases by hand with two streams, lib/align.cu:63-68,177-385):
//   uploader thread : packed offsets + metadata rebasing + H2D into input buffer i % 2
//   this thread     : wfagpu_amd_align_device (all kernels), then D2H into pinned staging i % 2
//   scatter thread  : staging -> the caller's wfa_alignment_result_t records (+ the -c check)
int run_shard(const CallArgs& a, Shard& sh) {
  if (sh.from >= sh.to) return 0;
  const bool timing = getenv("WFAGPU_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  if (sh.slot < 0 || sh.slot >= MAX_DEV) return -1;
  std::lock_guard<std::mutex> guard(g_dev_mu[sh.slot]);
  DevState* dp = nullptr;
  if (acquire_dev(sh.slot, sh.device, &dp)) return -1;
  DevState& d = *dp;
  HIP_OK(hipSetDevice(sh.device));
  const double t_created = now();

  const size_t n_all = sh.to - sh.from;
  size_t bs = a.opt.batch_size ? std::min<size_t>(a.opt.batch_size, n_all) : n_all;
  bs = std::max<size_t>(1, bs);
  // a single huge batch is cut in eight so that the stages have something to overlap (1M x 1 kbp pairs, two slices:
  // 66 ms host to host with four batches per slice, 59 ms with eight)
  static const size_t cut = getenv("WFAGPU_SLICE_BATCHES") ? (size_t)std::max(1, atoi(getenv("WFAGPU_SLICE_BATCHES"))) : 8;
  if (bs == n_all && n_all >= ((size_t)1 << 17)) bs = (n_all + cut - 1) / cut;
  std::vector<BatchPlan> plan;
  for (size_t from = sh.from; from < sh.to; from += bs) { BatchPlan b{}; b.from = from; b.to = std::min(sh.to, from + bs); plan.push_back(b); }
  const int nb = (int)plan.size();

  std::mutex mu;
  std::condition_variable cv;
  int uploaded = 0, computed = 0, scattered = 0;   // batches that finished each stage
  std::atomic<int> rc{0};
  double t_up = 0, t_dev = 0, t_d2h = 0, t_scatter = 0;
  auto advance = [&](int& counter) { { std::lock_guard<std::mutex> l(mu); ++counter; } cv.notify_all(); };
  auto wait_for = [&](const int& counter, int at_least) {
    std::unique_lock<std::mutex> l(mu);
    cv.wait(l, [&] { return counter >= at_least || rc.load() != 0; });
    return rc.load() == 0;
  };
  // (the flag changes under the mutex the waiters test it under: no wake-up can fall between their test and their block)
  auto fail = [&](int code) { { std::lock_guard<std::mutex> l(mu); rc.store(code); } cv.notify_all(); };

  std::thread uploader([&] {
    if (hipSetDevice(sh.device) != hipSuccess) { fail(-1); return; }
    std::vector<sequence_pair_t> hm;
    for (int i = 0; i < nb; ++i) {
      if (!wait_for(computed, i - 1)) return;           // input buffer i % 2 is free once batch i-2 has been computed
      const double t0 = now();
      BatchPlan& b = plan[i];
      const size_t n = b.to - b.from;
      // span of the batch inside the caller's buffer (lib/align.cu:80-93 takes it from the first/last
      // record; scanning is robust to any record order)
      size_t lo = SIZE_MAX, hi = 0;
      unsigned max_len = 0;
      for (size_t j = b.from; j < b.to; ++j) {
        const sequence_pair_t& m = a.meta[j];
        lo = std::min(lo, std::min(m.pattern_offset, m.text_offset));
        hi = std::max(hi, std::max(m.pattern_offset + m.pattern_len, m.text_offset + m.text_len));
        max_len = std::max(max_len, std::max(m.pattern_len, m.text_len));
      }
      lo &= ~(size_t)3;
      hi = std::min(a.seq_bytes, (hi + 4) & ~(size_t)3);
      b.lo = lo; b.span = hi - lo; b.max_len = max_len;
      // packed offsets: written into the caller's metadata like the reference
      // (lib/align.cu:103-115,363-377), relative to the batch
      b.packed_bytes = wfagpu_amd_fill_packed_offsets(a.meta + b.from, n);
      hm.assign(a.meta + b.from, a.meta + b.to);
      for (auto& m : hm) { m.pattern_offset -= lo; m.text_offset -= lo; }
      DevState::In& in = d.in[i & 1];
      auto ok = [&](hipError_t e, const char* what) { if (e != hipSuccess) { LOG_ERROR("HIP call %s failed: %s", what, hipGetErrorString(e)); fail(-1); return false; } return true; };
      if (b.span + 16 > in.seq_cap) {
        if (in.d_seq) (void)hipFree(in.d_seq);
        in.d_seq = nullptr; in.seq_cap = b.span + b.span / 8 + 16;
        if (!ok(hipMalloc(&in.d_seq, in.seq_cap), "hipMalloc(sequences)")) return;
      }
      if (n > in.meta_cap) {
        if (in.d_meta) (void)hipFree(in.d_meta);
        in.d_meta = nullptr; in.meta_cap = n + n / 8;
        if (!ok(hipMalloc(&in.d_meta, in.meta_cap * sizeof(sequence_pair_t)), "hipMalloc(metadata)")) return;
      }
      if (!ok(hipMemcpyAsync(in.d_seq, a.seq + lo, b.span, hipMemcpyHostToDevice, d.up), "H2D sequences")) return;
      if (!ok(hipMemcpyAsync(in.d_meta, hm.data(), n * sizeof(sequence_pair_t), hipMemcpyHostToDevice, d.up), "H2D metadata")) return;
      if (!ok(hipStreamSynchronize(d.up), "H2D sync")) return;
      t_up += now() - t0;
      advance(uploaded);
    }
  });

  std::thread scatterer([&] {
    for (int i = 0; i < nb; ++i) {
      if (!wait_for(computed, i + 1)) return;
      const double t0 = now();
      const BatchPlan& b = plan[i];
      const size_t n = b.to - b.from;
      const DevState::Out& o = d.out[i & 1];
      std::atomic<int> bad{0};
      auto work = [&](size_t j0, size_t j1) {
        for (size_t j = j0; j < j1; ++j) {
          wfa_alignment_result_t& r = a.results[b.from + j];
          r.error = (unsigned int)o.score[j];
          if (!a.cigar) continue;
          if (o.len[j] == 0xFFFFFFFFu) { LOG_ERROR("CIGAR recovery failed for pair %zu", b.from + j); bad.store(1); continue; }
          wfa_cigar_t& cg = r.cigar;
          const size_t need = (size_t)o.len[j] + 1;
          if (need > cg.buffer_size || !cg.buffer) {
            char* nbuf = static_cast<char*>(realloc(cg.buffer, need));
            if (!nbuf) { LOG_ERROR("Can not realloc CIGAR buffer"); exit(-1); }   // utils/wfa_cpu.c:77-80
            cg.buffer = nbuf; cg.buffer_size = need;
          }
          memcpy(cg.buffer, o.text + o.off[j], need);
          cg.last_free_position = o.len[j];
        }
      };
      const unsigned nt = a.cigar ? (unsigned)std::min<size_t>(std::max(1u, std::min(8u, sh.host_threads)), (n + 8191) / 8192) : 1u;
      if (nt <= 1) work(0, n);
      else {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        work(0, n / nt);
        for (auto& t : pool) t.join();
      }
      if (bad.load()) { fail(-1); return; }
      if (a.check) check_batch(a, b.from, b.to, i, sh.host_threads);
      t_scatter += now() - t0;
      advance(scattered);
    }
  });

  auto compute = [&]() -> int {
    for (int i = 0; i < nb; ++i) {
      if (!wait_for(uploaded, i + 1)) return rc.load();
      double t0 = now();
      BatchPlan& b = plan[i];
      const size_t n = b.to - b.from;
      DevState::In& in = d.in[i & 1];
      if (n > d.scores_cap) {
        if (d.d_scores) (void)hipFree(d.d_scores);
        d.d_scores = nullptr; d.scores_cap = n + n / 8;
        HIP_OK(hipMalloc(&d.d_scores, d.scores_cap * sizeof(int32_t)));
      }
      wfagpu_amd_batch_t wb{};
      wb.d_sequences = in.d_seq; wb.sequences_bytes = b.span; wb.d_metadata = in.d_meta; wb.num_pairs = n;
      wb.packed_bytes = b.packed_bytes; wb.max_seq_len = b.max_len;
      const char* d_text = nullptr; const unsigned long long* d_off = nullptr; const unsigned int* d_len = nullptr;
      wfagpu_amd_hint_same_stream(d.ctx, i > 0 ? 1 : 0);     // the batches of a call come from one stream of reads
      const int arc = wfagpu_amd_align_device(d.ctx, &wb, a.opt.penalties, a.opt.max_error, a.opt.band, a.opt.threads_per_block, a.cigar,
                                              d.d_scores, &d_text, &d_off, &d_len);
      if (arc) return arc;
      t_dev += now() - t0; t0 = now();
      if (!wait_for(scattered, i - 1)) return rc.load();   // staging i % 2 is free once batch i-2 has been scattered
      DevState::Out& o = d.out[i & 1];
      if (n > o.n_cap) {
        const size_t cap = n + n / 8;
        if (grow_pinned(&o.score, cap)) return -1;
        o.n_cap = cap;
      }
      if (a.cigar && n > o.cig_cap) {
        const size_t cap = n + n / 8;
        if (grow_pinned(&o.off, cap) || grow_pinned(&o.len, cap)) return -1;
        o.cig_cap = cap;
      }
      HIP_OK(hipMemcpyAsync(o.score, d.d_scores, n * sizeof(int32_t), hipMemcpyDeviceToHost, d.down));
      if (a.cigar) {
        wfagpu_amd_stats_t stt; wfagpu_amd_last_stats(d.ctx, &stt);
        b.text_bytes = stt.text_bytes;
        if (stt.text_bytes + 1 > o.text_cap) {
          const size_t cap = (size_t)stt.text_bytes + (size_t)stt.text_bytes / 8 + 4096;
          if (grow_pinned(&o.text, cap)) return -1;
          o.text_cap = cap;
        }
        HIP_OK(hipMemcpyAsync(o.off, d_off, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, d.down));
        HIP_OK(hipMemcpyAsync(o.len, d_len, n * sizeof(unsigned int), hipMemcpyDeviceToHost, d.down));
        if (stt.text_bytes) HIP_OK(hipMemcpyAsync(o.text, d_text, stt.text_bytes, hipMemcpyDeviceToHost, d.down));
      }
      HIP_OK(hipStreamSynchronize(d.down));
      t_d2h += now() - t0;
      advance(computed);
    }
    return 0;
  };
  const int crc = compute();
  if (crc) fail(crc);
  uploader.join();
  scatterer.join();
  if (timing)
    fprintf(stderr, "[wfagpu timing] device %d: %d batches, acquire %.1f ms, upload %.1f, device %.1f, d2h %.1f, scatter %.1f (overlapped), total %.1f\n",
            sh.device, nb, t_created - t_begin, t_up, t_dev, t_d2h, t_scatter, now() - t_begin);
  return rc.load();
}

void launch_impl(char* seq, size_t seq_bytes, sequence_pair_t* meta, wfa_alignment_result_t* results,
                 wfa_alignment_options_t opt, bool check, bool cigar) {
  if (!seq || !meta || !results) { LOG_ERROR("Invalid buffers."); return; }
  const size_t n = opt.num_alignments;
  if (n == 0) return;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    LOG_ERROR("No HIP device available.");
    exit(-1);   // utils/cuda_utils.cuh:28-35: device errors are fatal
  }
  const int want = g_num_devices.load();
  if (want > 0) ndev = std::min(ndev, want);
  // WFAGPU_VIRTUAL_DEVICES=k (tests): k shards with their own threads, contexts and streams, mapped round-robin
  // onto the physical devices -- exercises the multi-device path on a single GPU
  const int physical = ndev;
  // Two slices per device for big calls: the kernels of one slice fill the gaps (host round trips, backtrace
  // tails, copies) of the other -- 1M 1 kbp pairs with CIGARs: 81 -> 73 ms host to host, same cold-call time.
  if (n / (size_t)physical >= ((size_t)1 << 18)) ndev = std::min(MAX_DEV, 2 * physical);
  if (const char* e = getenv("WFAGPU_VIRTUAL_DEVICES")) ndev = std::max(1, std::min(MAX_DEV, atoi(e)));
  ndev = (int)std::min<size_t>((size_t)ndev, n);
  CallArgs a{seq, seq_bytes, meta, results, opt, check, cigar};
  g_check_failures.store(0);
  std::vector<Shard> shards(ndev);
  // Contiguous slices of equal WORK, not of equal count: the wavefront work of a pair grows with P x T (score^2 at a
  // given error rate), so a ragged batch cut by count would leave devices idle.  One pass over the records.
  std::vector<size_t> cut(ndev + 1, n);
  cut[0] = 0;
  if (ndev > 1) {
    long double total = 0;
    for (size_t i = 0; i < n; ++i) total += (long double)meta[i].pattern_len * meta[i].text_len + 1024.0L;   // (+ a per-pair constant)
    long double acc = 0;
    int d = 1;
    for (size_t i = 0; i < n && d < ndev; ++i) {
      acc += (long double)meta[i].pattern_len * meta[i].text_len + 1024.0L;
      while (d < ndev && acc >= total * d / ndev) cut[d++] = i + 1;
    }
  }
  const unsigned host_threads = usable_host_threads();
  for (int d = 0; d < ndev; ++d) {
    shards[d].device = d % physical;
    shards[d].slot = d;
    shards[d].from = cut[d];
    shards[d].to = cut[d + 1];
    shards[d].host_threads = std::max(1u, std::min(64u, host_threads / (unsigned)ndev));
  }
  if (ndev == 1) {
    shards[0].rc = run_shard(a, shards[0]);
  } else {
    std::vector<std::thread> th;
    // (own threads, one per slice: each may be moved to the cores next to its GPU; the caller's thread is never touched)
    for (int d = 0; d < ndev; ++d)
      th.emplace_back([&a, &shards, d, physical] {
        if (physical > 1 || getenv("WFAGPU_FORCE_NUMA_PIN")) pin_to_device_node(shards[d].device);    // (the variable: test hook for one-GPU boxes)
        shards[d].rc = run_shard(a, shards[d]);
      });
    for (auto& t : th) t.join();
  }
  for (auto& s : shards)
    if (s.rc) { LOG_ERROR("Alignment failed on device %d (code %d).", s.device, s.rc); exit(-1); }
}

}  // namespace

extern "C" {

void wfagpu_amd_set_num_devices(int n) { g_num_devices.store(n < 0 ? 0 : n); }

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

void get_num_cuda_devices(int* n) {
  if (!n) return;
  if (hipGetDeviceCount(n) != hipSuccess) *n = 0;
}

char* get_cuda_dev_name(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return strdup("unknown");
  return strdup(prop.name);
}

int get_cuda_SM_count(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return prop.multiProcessorCount;
}

void get_cuda_capability(int dev, int* major, int* minor) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { if (major) *major = 0; if (minor) *minor = 0; return; }
  if (major) *major = prop.major;
  if (minor) *minor = prop.minor;
}

}  // extern "C"
