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
#include <thread>
#include <vector>

#include "../../include/wfa_gpu_device.h"
#include "../utils/logger.h"
#include "../utils/verification.h"

namespace {

std::atomic<int> g_num_devices{0};   // 0: all visible

struct Shard {
  int device;
  size_t from, to;   // [from, to)
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

int check_batch(const CallArgs& a, size_t from, size_t to, int batch_idx) {
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
  unsigned nt = std::max(1u, std::thread::hardware_concurrency());
  nt = (unsigned)std::min<size_t>(nt, (to - from + 15) / 16);
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  fprintf(stderr, "(Batch %d) correct=%ld Incorrect=%ld Average score=%f\n", batch_idx, correct.load(), incorrect.load(),
          (to > from) ? (double)sum.load() / (double)(to - from) : 0.0);
  return incorrect.load() ? 1 : 0;
}

// One pipeline lane: its own context (stream, device buffers) working through every `stride`-th batch of
// the shard.  Two lanes per device give the overlap the reference builds by hand with two streams
// (lib/align.cu:63-68,177-385): while one lane's kernels run, the other lane uploads its next batch or
// scatters its previous results on the host.
int run_lane(const CallArgs& a, const Shard& sh, size_t bs, size_t first, size_t stride) {
  wfagpu_amd_ctx_t* ctx = nullptr;
  wfagpu_amd_config_t cfg{};
  cfg.device = sh.device;
  // one-shot call: allocating device memory costs ~33 ms per GiB, so keep the backtrace arena small and
  // let big batches run in several passes
  cfg.arena_limit_bytes = (size_t)4 << 30;
  if (wfagpu_amd_create(&ctx, &cfg)) return -1;
  HIP_OK(hipSetDevice(sh.device));
  char* d_seq = nullptr; size_t d_seq_cap = 0;
  sequence_pair_t* d_meta = nullptr; size_t d_meta_cap = 0;
  int32_t* d_scores = nullptr; size_t d_scores_cap = 0;
  std::vector<sequence_pair_t> hm;
  std::vector<int32_t> hs;
  std::vector<unsigned long long> hoff;
  std::vector<unsigned int> hlen;
  std::vector<char> htext;
  int rc = 0;
  for (size_t batch_idx = first; sh.from + batch_idx * bs < sh.to && rc == 0; batch_idx += stride) {
    const size_t from = sh.from + batch_idx * bs;
    const size_t to = std::min(sh.to, from + bs);
    const size_t n = to - from;
    // span of the batch inside the caller's buffer (lib/align.cu:80-93 takes
    // it from the first/last record; scanning is robust to any record order)
    size_t lo = SIZE_MAX, hi = 0;
    unsigned max_len = 0;
    for (size_t i = from; i < to; ++i) {
      const sequence_pair_t& m = a.meta[i];
      lo = std::min(lo, std::min(m.pattern_offset, m.text_offset));
      hi = std::max(hi, std::max(m.pattern_offset + m.pattern_len, m.text_offset + m.text_len));
      max_len = std::max(max_len, std::max(m.pattern_len, m.text_len));
    }
    lo &= ~(size_t)3;
    hi = std::min(a.seq_bytes, (hi + 4) & ~(size_t)3);
    const size_t span = hi - lo;
    // packed offsets: written into the caller's metadata like the reference
    // (lib/align.cu:103-115,363-377), relative to the batch
    const size_t packed_bytes = wfagpu_amd_fill_packed_offsets(a.meta + from, n);
    hm.assign(a.meta + from, a.meta + to);
    for (auto& m : hm) { m.pattern_offset -= lo; m.text_offset -= lo; }
    if (span + 16 > d_seq_cap) { if (d_seq) hipFree(d_seq); d_seq_cap = span + span / 2 + 16; HIP_OK(hipMalloc(&d_seq, d_seq_cap)); }
    if (n > d_meta_cap) { if (d_meta) hipFree(d_meta); d_meta_cap = n + n / 2; HIP_OK(hipMalloc(&d_meta, d_meta_cap * sizeof(sequence_pair_t))); }
    if (n > d_scores_cap) { if (d_scores) hipFree(d_scores); d_scores_cap = n + n / 2; HIP_OK(hipMalloc(&d_scores, d_scores_cap * sizeof(int32_t))); }
    HIP_OK(hipMemcpy(d_seq, a.seq + lo, span, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_meta, hm.data(), n * sizeof(sequence_pair_t), hipMemcpyHostToDevice));
    wfagpu_amd_batch_t b{};
    b.d_sequences = d_seq; b.sequences_bytes = span; b.d_metadata = d_meta; b.num_pairs = n;
    b.packed_bytes = packed_bytes; b.max_seq_len = max_len;
    const char* d_text = nullptr; const unsigned long long* d_off = nullptr; const unsigned int* d_len = nullptr;
    const int arc = wfagpu_amd_align_device(ctx, &b, a.opt.penalties, a.opt.max_error, a.opt.band, a.opt.threads_per_block, a.cigar, d_scores,
                                            &d_text, &d_off, &d_len);
    if (arc) { rc = arc; break; }
    hs.resize(n);
    HIP_OK(hipMemcpy(hs.data(), d_scores, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) a.results[from + i].error = (unsigned int)hs[i];
    if (a.cigar) {
      wfagpu_amd_stats_t stt; wfagpu_amd_last_stats(ctx, &stt);
      hoff.resize(n); hlen.resize(n); htext.resize(stt.text_bytes + 1);
      HIP_OK(hipMemcpy(hoff.data(), d_off, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(hlen.data(), d_len, n * sizeof(unsigned int), hipMemcpyDeviceToHost));
      if (stt.text_bytes) HIP_OK(hipMemcpy(htext.data(), d_text, stt.text_bytes, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) {
        wfa_cigar_t& cg = a.results[from + i].cigar;
        if (hlen[i] == 0xFFFFFFFFu) { LOG_ERROR("CIGAR recovery failed for pair %zu", from + i); rc = -1; continue; }
        const size_t need = (size_t)hlen[i] + 1;
        if (need > cg.buffer_size || !cg.buffer) {
          char* nb = static_cast<char*>(realloc(cg.buffer, need));
          if (!nb) { LOG_ERROR("Can not realloc CIGAR buffer"); exit(-1); }   // utils/wfa_cpu.c:77-80
          cg.buffer = nb; cg.buffer_size = need;
        }
        memcpy(cg.buffer, htext.data() + hoff[i], need);
        cg.last_free_position = hlen[i];
      }
    }
    if (a.check && rc == 0) check_batch(a, from, to, (int)batch_idx);
  }
  if (d_seq) hipFree(d_seq);
  if (d_meta) hipFree(d_meta);
  if (d_scores) hipFree(d_scores);
  wfagpu_amd_destroy(ctx);
  return rc;
}

int run_shard(const CallArgs& a, Shard& sh) {
  if (sh.from >= sh.to) return 0;
  const size_t n = sh.to - sh.from;
  size_t bs = a.opt.batch_size ? std::min<size_t>(a.opt.batch_size, n) : n;
  bs = std::max<size_t>(1, bs);
  // a single huge batch is cut in four so that the two lanes have something to overlap
  if (bs == n && n >= ((size_t)1 << 18) && !a.cigar) bs = (n + 3) / 4;
  const size_t nbatches = (n + bs - 1) / bs;
  // CIGAR mode allocates a backtrace arena per lane, and device allocation (~33 ms per GiB) is not
  // something a second lane can hide: measured 0.27 s (one lane) vs 0.39 s (two) per 1M 1 kbp pairs;
  // score-only runs gain from the overlap (0.174 -> 0.156 s)
  if (nbatches < 2 || a.cigar) return run_lane(a, sh, bs, 0, 1);
  int rc[2] = {0, 0};
  std::thread other([&] { rc[1] = run_lane(a, sh, bs, 1, 2); });
  rc[0] = run_lane(a, sh, bs, 0, 2);
  other.join();
  return rc[0] ? rc[0] : rc[1];
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
  ndev = (int)std::min<size_t>((size_t)ndev, n);
  CallArgs a{seq, seq_bytes, meta, results, opt, check, cigar};
  std::vector<Shard> shards(ndev);
  for (int d = 0; d < ndev; ++d) {
    shards[d].device = d;
    shards[d].from = n * d / ndev;
    shards[d].to = n * (d + 1) / ndev;
  }
  if (ndev == 1) {
    shards[0].rc = run_shard(a, shards[0]);
  } else {
    std::vector<std::thread> th;
    for (int d = 0; d < ndev; ++d) th.emplace_back([&a, &shards, d] { shards[d].rc = run_shard(a, shards[d]); });
    for (auto& t : th) t.join();
  }
  for (auto& s : shards)
    if (s.rc) { LOG_ERROR("Alignment failed on device %d (code %d).", s.device, s.rc); exit(-1); }
}

}  // namespace

extern "C" {

void wfagpu_amd_set_num_devices(int n) { g_num_devices.store(n < 0 ? 0 : n); }

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
