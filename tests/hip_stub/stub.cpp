// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): streams as worker threads, events, host-backed "device" memory,
// and a wfagpu_amd context whose align call is the ORACLE (oracle/wfa_oracle.c) -- the launch pipeline is what is under test.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/wfa_gpu_device.h"
#include "../../oracle/wfa_oracle.h"

struct StubEvent {
  std::mutex mu; std::condition_variable cv;
  unsigned long long recorded = 0, done = 0;      // generation counters
};
struct StubStream {
  std::mutex mu; std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false; unsigned long long issued = 0, retired = 0;
  std::thread worker;
  StubStream() : worker([this] { run(); }) {}
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        f = std::move(q.front()); q.pop_front();
      }
      f();
      { std::lock_guard<std::mutex> l(mu); ++retired; cv.notify_all(); }
    }
  }
  void push(std::function<void()> f) { { std::lock_guard<std::mutex> l(mu); q.push_back(std::move(f)); ++issued; } cv.notify_all(); }
  void drain() { std::unique_lock<std::mutex> l(mu); const unsigned long long want = issued; cv.wait(l, [&] { return retired >= want; }); }
  ~StubStream() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_all(); worker.join(); }
};

static std::atomic<int> g_devices{1};
void stub_set_device_count(int n) { g_devices.store(n); }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "stub error"; }
hipError_t hipGetDeviceCount(int* n) { *n = g_devices.load(); return hipSuccess; }
static thread_local int t_device = 0;      // (the current device is per thread, like the runtime's)
hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_devices.load()) return hipErrorInvalidValue; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "stub"); p->multiProcessorCount = 256; p->major = 9; p->minor = 5; p->sharedMemPerBlock = 160 << 10; return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char*, int, int) { return hipErrorInvalidValue; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new StubStream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new StubStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { s->drain(); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
  s->push([=] { memcpy(dst, src, n); });
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new StubEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  unsigned long long gen;
  { std::lock_guard<std::mutex> l(e->mu); gen = ++e->recorded; }
  // (notified under the mutex: the waiter may destroy the event the moment it has seen it done)
  s->push([e, gen] { std::lock_guard<std::mutex> l(e->mu); if (e->done < gen) e->done = gen; e->cv.notify_all(); });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> l(e->mu);
  const unsigned long long want = e->recorded;
  e->cv.wait(l, [&] { return e->done >= want; });
  return hipSuccess;
}

// ---- the context: what csrc/wfa_host.hip is to the launch layer, computed by the oracle ------------------------------------
struct wfagpu_amd_ctx {
  hipStream_t stream = nullptr;
  std::vector<char> text[2]; std::vector<unsigned long long> off[2]; std::vector<unsigned int> len[2];
  int out_set = 0;
  wfagpu_amd_stats_t stats{};
};
extern "C" {
int wfagpu_amd_create(wfagpu_amd_ctx_t** out, const wfagpu_amd_config_t*) { auto* c = new wfagpu_amd_ctx(); c->stream = new StubStream(); *out = c; return 0; }
void wfagpu_amd_destroy(wfagpu_amd_ctx_t* c) { if (c) { delete c->stream; delete c; } }
void wfagpu_amd_trim(wfagpu_amd_ctx_t*) {}
int wfagpu_amd_prime(wfagpu_amd_ctx_t*) { return 0; }
void wfagpu_amd_hint_same_stream(wfagpu_amd_ctx_t*, int) {}
void wfagpu_amd_last_stats(const wfagpu_amd_ctx_t* c, wfagpu_amd_stats_t* out) { *out = c->stats; }
size_t wfagpu_amd_fill_packed_offsets(sequence_pair_t* m, size_t n) {
  size_t o = 0;
  for (size_t i = 0; i < n; ++i) { m[i].pattern_offset_packed = o; o += 4 * (((size_t)m[i].pattern_len + 15) / 16 + 1); m[i].text_offset_packed = o; o += 4 * (((size_t)m[i].text_len + 15) / 16 + 1); }
  return o;
}
int wfagpu_amd_align_device(wfagpu_amd_ctx_t* c, const wfagpu_amd_batch_t* b, affine_penalties_t pen, int, int, int, bool cigar, int32_t* d_scores,
                            const char** d_text, const unsigned long long** d_off, const unsigned int** d_len) {
  const size_t n = b->num_pairs;
  const int set = c->out_set;
  auto& text = c->text[set]; auto& off = c->off[set]; auto& len = c->len[set];
  text.clear(); off.assign(n, 0); len.assign(n, 0);
  oracle_aligner_t* al = oracle_aligner_new(pen.x, pen.o, pen.e);
  std::vector<char> p, t, cg;
  static const char lut[4] = {'A', 'C', 'T', 'G'};
  for (size_t i = 0; i < n; ++i) {
    const sequence_pair_t& m = b->d_metadata[i];
    const char *pp, *tp;
    if (b->d_packed) {      // 2-bit words, first base in the low bits (the host packer's layout)
      const uint32_t* w = static_cast<const uint32_t*>(b->d_packed);
      p.resize(m.pattern_len + 1); t.resize(m.text_len + 1);
      for (unsigned j = 0; j < m.pattern_len; ++j) p[j] = lut[(w[m.pattern_offset_packed / 4 + j / 16] >> (2 * (j % 16))) & 3];
      for (unsigned j = 0; j < m.text_len; ++j) t[j] = lut[(w[m.text_offset_packed / 4 + j / 16] >> (2 * (j % 16))) & 3];
      pp = p.data(); tp = t.data();
    } else {
      pp = b->d_sequences + m.pattern_offset; tp = b->d_sequences + m.text_offset;
    }
    if (cigar) {
      cg.resize(2 * ((size_t)m.pattern_len + m.text_len) + 16);
      d_scores[i] = oracle_align(al, pp, (int)m.pattern_len, tp, (int)m.text_len, cg.data(), cg.size(), nullptr);
      off[i] = text.size(); len[i] = (unsigned)strlen(cg.data());
      text.insert(text.end(), cg.data(), cg.data() + len[i] + 1);
    } else {
      d_scores[i] = oracle_score(al, pp, (int)m.pattern_len, tp, (int)m.text_len, 0, nullptr);
    }
  }
  oracle_aligner_delete(al);
  c->stats = wfagpu_amd_stats_t{};
  c->stats.text_bytes = text.size();
  if (cigar) { *d_text = text.data(); *d_off = off.data(); *d_len = len.data(); c->out_set ^= 1; }
  return 0;
}
}
