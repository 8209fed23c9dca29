// Host side of the device-resident entry points (include/wfa_gpu_device.h):
// context, buffer management, the tier-escalation driver and timing.
//
// Replaces the reference's launch/memory layer (lib/sequence_alignment.cu:31-470,
// lib/sequence_packing.cu:96-116) and the per-batch body of its orchestrator
// (lib/align.cu:177-385).  Differences that matter:
//   * a pair the first kernel tier cannot finish (score past max_error, or a
//     wavefront wider than the tier's LDS ring) is re-queued ON THE DEVICE
//     into a wider tier; the reference recomputed it on the CPU
//     (utils/wfa_cpu.c:59-84).  There is no CPU compute path in this library.
//   * backtrace memory is a bump arena shared by all workgroups; if a batch
//     outgrows it the unfinished pairs run in a further pass after the
//     finished ones have been traced (the reference sized per-worker arenas
//     from max_error^2 and aborted beyond 2^32 elements,
//     lib/sequence_alignment.cu:36-42).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/wfa_gpu_device.h"
#include "wfa_device.h"

static_assert(sizeof(sequence_pair_t) == sizeof(WfaSeqPair), "ABI mirror");

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      fprintf(stderr, "[!] ERROR: HIP %s failed: %s (%s:%d)\n", #expr,                  \
              hipGetErrorString(_e), __FILE__, __LINE__);                               \
      return -1;                                                                        \
    }                                                                                   \
  } while (0)

// helper-kernel launch + launch-error check (the align/trace launchers are checked at their call sites)
#define LAUNCH_K(kern, grid, block, lds, stream, ...)                 \
  do {                                                                 \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);   \
    HIP_TRY(hipGetLastError());                                        \
  } while (0)

// the same with the END of the kernel stamped into `ev1` by its own dispatch packet (wfa_launch_timed: no barrier packet of a hipEventRecord)
#define LAUNCH_K_END(kern, grid, block, lds, stream, ev1, ...)                                       \
  do {                                                                                                 \
    wfa_launch_timed(kern, grid, block, lds, stream, (hipEvent_t) nullptr, ev1, __VA_ARGS__);          \
    HIP_TRY(hipGetLastError());                                                                        \
  } while (0)

namespace {

// hipFree waits for the whole device to go idle: inside a call -- other lanes' kernels running -- a buffer that has to grow
// would stall its lane for 4-8 ms per free (a cold call grows a dozen buffers: first the sample's sizes, then the batch's).
// Buffers that are outgrown during a call are RETIRED instead (the context's list, named by this thread-local while a call
// runs) and freed by wfagpu_amd_trim / wfagpu_amd_destroy, or at the start of a later call once they add up to 1 GiB.
struct Retired { void* p; size_t bytes; };
thread_local std::vector<Retired>* t_retire = nullptr;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes, hipStream_t stream, size_t preserve = 0) {
    if (bytes <= cap) return 0;
    size_t ncap = std::max(bytes, cap + cap / 2);
    void* np = nullptr;
    if (hipMalloc(&np, ncap) != hipSuccess) {
      // retry with the exact size
      ncap = bytes;
      hipError_t e = hipMalloc(&np, ncap);
      if (e != hipSuccess) {
        fprintf(stderr, "[!] ERROR: hipMalloc(%zu) failed: %s\n", ncap, hipGetErrorString(e));
        return -1;
      }
    }
    if (p) {
      // (what the old buffer holds -- CIGAR text of earlier passes -- must survive the move: a failed copy is an error)
      hipError_t e = preserve ? hipMemcpyAsync(np, p, preserve, hipMemcpyDeviceToDevice, stream) : hipSuccess;
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) {
        fprintf(stderr, "[!] ERROR: moving a device buffer of %zu bytes failed: %s\n", preserve, hipGetErrorString(e));
        hipFree(np);
        return -1;
      }
      // (big ones -- a backtrace arena that grows between passes -- go back at once: their memory is needed)
      if (t_retire && cap < ((size_t)256 << 20)) t_retire->push_back({p, cap}); else hipFree(p);
    }
    p = np; cap = ncap;
    return 0;
  }
  void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

// device counters (u64 slots)
// (CT_LCELLS/CT_LIST and CT_LCELLS2/CT_LIST2: cell count and failure-list length of the first and second wavefront launch of
// a chain, zeroed in pairs; CT_NOMEM: length of the list of pairs that ran out of arena in the current pass; CT_UNFIN: pairs of the
// batch that are not DONE, counted just before every chain's synchronisation)
enum { CT_ARENA = 0, CT_OPS = 1, CT_TEXT = 2, CT_LCELLS = 3, CT_LIST = 4, CT_SUM_OPS = 5, CT_SUM_TEXT = 6, CT_CELLS = 7, CT_MAX_SCORE = 8, CT_NRAW = 9, CT_SCRATCH = 10,
       CT_LCELLS2 = 11, CT_LIST2 = 12, CT_NOMEM = 13, CT_UNFIN = 14, CT_N = 15 };
constexpr size_t CT_BYTES = 128;      // the u64 slots, padded to a cache line boundary (CT_N * 8 = 120)
static_assert(CT_N * sizeof(unsigned long long) <= CT_BYTES && CT_N <= 32, "counter block");
// Behind the u64 slots and the eight claim counters: tier 5's partial sums, four u64 per wavefront of a launch (WfaAlignParams::wave_parts;
// a launch has at most 32 wavefronts per CU).  On the device when a kernel behind the launch adds them up; in the pinned host copy
// of this block (the kernel stores them there itself) when nothing runs behind the launch.
constexpr size_t CT_PARTS_OFF = CT_BYTES + 8 * 64;
constexpr size_t CT_PARTS_PER_CU = 32;
constexpr size_t CT_PART_BYTES = 32;

constexpr uint32_t MASK(uint32_t st) { return 1u << st; }

// Appends the elements of a block that carry `take` to a list: ONE atomic per workgroup (a counter word serves ~88
// claims per microsecond: with one per wavefront a 1M-pair list took 190 us), order kept within the block.
__device__ __forceinline__ void block_append(bool take, uint32_t value, uint32_t* __restrict__ out, unsigned long long* __restrict__ out_count) {
  __shared__ uint32_t wave_n[16];
  __shared__ unsigned long long block_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const unsigned long long bal = __ballot(take);
  if (lane == 0) wave_n[wave] = (uint32_t)__builtin_popcountll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < nw; ++w) { const uint32_t c = wave_n[w]; wave_n[w] = tot; tot += c; }
    block_base = tot ? atomicAdd(out_count, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  if (take) out[block_base + wave_n[wave] + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = value;
}

// Tier 5 leaves the cells of a launch as one partial sum per wavefront (plain stores): the kernel behind the launch adds them up.
// (One atomic per wavefront on the call's counter line -- ~11 ns each whoever issues it, 3571 wavefronts ending together -- was 15 of the
// 107 us of a launch of 100k configs[1] pairs.)
__device__ __forceinline__ void add_wave_parts(const unsigned long long* __restrict__ parts, uint32_t n_parts, unsigned long long* __restrict__ out) {
  unsigned long long s = 0;
  for (uint32_t i = threadIdx.x; i < n_parts; i += blockDim.x) s += parts[4 * i];      // ({cells, ...} per wavefront: WfaAlignParams::wave_parts)
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// pairs of `work` whose status is in `mask` -> out list (appended at *out_count).  n_dev (optional): the real length of
// `work` where only the device knows it yet; n is then an upper bound.
// work_ctr (optional): the eight claim counters of the wavefront launch that has just finished, zeroed here for the next
// one (a hipMemsetAsync per launch is a kernel launch of its own: ~5 us each, a dozen of them were 15 % of a 100k-pair step).
__global__ void __launch_bounds__(1024) k_compact(const uint32_t* __restrict__ work, uint32_t n, const unsigned long long* __restrict__ n_dev,
                          const uint32_t* __restrict__ status,
                          uint32_t mask, uint32_t* __restrict__ out, unsigned long long* __restrict__ out_count,
                          unsigned int* __restrict__ work_ctr = nullptr,
                          const unsigned long long* __restrict__ parts = nullptr, uint32_t n_parts = 0, unsigned long long* __restrict__ parts_out = nullptr) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (work_ctr && blockIdx.x == 0 && threadIdx.x < 8) work_ctr[threadIdx.x * 16] = 0u;
  if (parts && blockIdx.x == 0) add_wave_parts(parts, n_parts, parts_out);
  if (n_dev) n = min(n, (uint32_t)*n_dev);
  bool take = false; uint32_t pair = 0;
  if (gid < n) { pair = work ? work[gid] : gid; take = (mask >> status[pair]) & 1u; }
  block_append(take, pair, out, out_count);
}

// Same, restricted to pairs whose longer sequence has len_lo <= length <= len_hi (length buckets).
__global__ void __launch_bounds__(1024) k_compact_len(uint32_t n, const uint32_t* __restrict__ status, uint32_t mask, const WfaSeqPair* __restrict__ meta,
                              uint32_t len_lo, uint32_t len_hi, uint32_t* __restrict__ out, unsigned long long* __restrict__ out_count) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  bool take = false;
  if (gid < n && ((mask >> status[gid]) & 1u)) {
    const uint32_t len = max(meta[gid].pattern_len, meta[gid].text_len);
    take = len >= len_lo && len <= len_hi;
  }
  block_append(take, gid, out, out_count);
}

// bounds for the trace scratch: sum over finished pairs of the op-list and
// text sizes; also the total of computed cells and the largest score.  Grid-stride, FOUR atomics per block, few blocks: the
// counters share one cache line and an atomic on it costs ~11 ns whoever issues it (one atomicMax per wavefront + three adds per
// block of a 391-block grid: 22 of the 250 microseconds of a BASELINE configs[1] step).
// (n_all > 0: the same launch also counts the pairs of the WHOLE batch that are not DONE -- what k_count_unfinished does -- into
// ct[CT_UNFIN]: the statuses are final once the chain's wavefront launches are through, the backtrace does not touch them; one
// launch and its gap fewer per chain: ~8 us of a 175 us step of 100k short reads)
__global__ void __launch_bounds__(256) k_trace_bounds(const uint32_t* __restrict__ work, uint32_t n, const uint32_t* __restrict__ status,
                               const int32_t* __restrict__ score, const uint32_t* __restrict__ cells, int min_op_cost, int item_chars,
                               unsigned long long* __restrict__ ct, uint32_t n_all) {
  __shared__ unsigned long long part[4][4];
  unsigned long long ops = 0, txt = 0, cl = 0;
  uint32_t smax = 0;
  if (n_all) {
    uint32_t bad = 0;
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < n_all; i += gridDim.x * 1024u) {
      if (i + 3u < n_all) {
        const uint4 v = *reinterpret_cast<const uint4*>(status + i);      // (the status array is 16-byte aligned, i a multiple of 4)
        bad += (v.x != WFA_ST_DONE) + (v.y != WFA_ST_DONE) + (v.z != WFA_ST_DONE) + (v.w != WFA_ST_DONE);
      } else {
        for (uint32_t j = i; j < n_all; ++j) bad += status[j] != WFA_ST_DONE;
      }
    }
    for (int d = 32; d > 0; d >>= 1) bad += __shfl_down(bad, d);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&ct[CT_UNFIN], (unsigned long long)bad);
  }
  for (uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x; gid < n; gid += gridDim.x * blockDim.x) {
    const uint32_t pair = work ? work[gid] : gid;
    if (status[pair] == WFA_ST_DONE) {
      const uint32_t s = (uint32_t)score[pair];
      smax = max(smax, s);
      ops += (s + 3u) & ~3u;
      const uint32_t m = s / (uint32_t)min_op_cost;
      txt += (unsigned long long)item_chars * (2ull * m + 1ull) + 1ull;
      if (cells) cl += cells[pair];       // (of the launch that finished the pair)
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    ops += __shfl_down((unsigned long long)ops, d);
    txt += __shfl_down((unsigned long long)txt, d);
    cl += __shfl_down((unsigned long long)cl, d);
    smax = max(smax, (uint32_t)__shfl_down((int)smax, d));
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { part[0][wv] = ops; part[1][wv] = txt; part[2][wv] = cl; part[3][wv] = smax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ops = part[0][0] + part[0][1] + part[0][2] + part[0][3];
    txt = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    cl = part[2][0] + part[2][1] + part[2][2] + part[2][3];
    const unsigned long long sm = max(max(part[3][0], part[3][1]), max(part[3][2], part[3][3]));
    if (sm) atomicMax(&ct[CT_MAX_SCORE], sm);
    if (ops) atomicAdd(&ct[CT_SUM_OPS], ops);
    if (txt) atomicAdd(&ct[CT_SUM_TEXT], txt);
    if (cl) atomicAdd(&ct[CT_CELLS], cl);
  }
}

// The tail of a chain with ONE wavefront launch, in one kernel: compaction of the launch's failures (k_compact), bounds of what
// finished (k_trace_bounds) and the count of unfinished pairs of the whole batch (k_count_unfinished).  Three launches of ~4 us
// each (+ the gaps between them) were a tenth of a 175 us step of 100k short reads.  One thread per list entry.
__global__ void __launch_bounds__(256) k_chain_tail(const uint32_t* __restrict__ work, uint32_t n, const uint32_t* __restrict__ status,
                             uint32_t mask, uint32_t* __restrict__ out, unsigned long long* __restrict__ out_count, unsigned int* __restrict__ work_ctr,
                             const int32_t* __restrict__ score, const uint32_t* __restrict__ cells, int min_op_cost, int item_chars,
                             unsigned long long* __restrict__ ct, uint32_t n_all,
                             const unsigned long long* __restrict__ parts, uint32_t n_parts, unsigned long long* __restrict__ parts_out) {
  // (few blocks, grid-stride: all the counters of a call share one cache line and an atomic on it costs ~11 ns whoever issues it --
  // one block per 64 list entries, four atomics each, made this kernel 70 us for 100k pairs)
  __shared__ unsigned long long part[4][4];
  if (work_ctr && blockIdx.x == 0 && threadIdx.x < 8) work_ctr[threadIdx.x * 16] = 0u;
  if (parts && blockIdx.x == gridDim.x - 1) add_wave_parts(parts, n_parts, parts_out);
  unsigned long long ops = 0, txt = 0, cl = 0;
  uint32_t smax = 0;
  for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {      // (uniform per block: block_append has barriers)
    const uint32_t gid = base + threadIdx.x;
    bool take = false; uint32_t pair = 0, st = WFA_ST_PENDING;
    if (gid < n) { pair = work ? work[gid] : gid; st = status[pair]; take = (mask >> st) & 1u; }
    block_append(take, pair, out, out_count);
    if (gid < n && st == WFA_ST_DONE) {
      const uint32_t sc = (uint32_t)score[pair];
      smax = max(smax, sc);
      ops += (sc + 3u) & ~3u;
      txt += (unsigned long long)item_chars * (2ull * (sc / (uint32_t)min_op_cost) + 1ull) + 1ull;
      if (cells) cl += cells[pair];
    }
    __syncthreads();      // (block_append's shared words are written again by the next round)
  }
  for (int d = 32; d > 0; d >>= 1) {
    ops += __shfl_down((unsigned long long)ops, d);
    txt += __shfl_down((unsigned long long)txt, d);
    cl += __shfl_down((unsigned long long)cl, d);
    smax = max(smax, (uint32_t)__shfl_down((int)smax, d));
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { part[0][wv] = ops; part[1][wv] = txt; part[2][wv] = cl; part[3][wv] = smax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ops = part[0][0] + part[0][1] + part[0][2] + part[0][3];
    txt = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    cl = part[2][0] + part[2][1] + part[2][2] + part[2][3];
    const unsigned long long sm = max(max(part[3][0], part[3][1]), max(part[3][2], part[3][3]));
    if (sm) atomicMax(&ct[CT_MAX_SCORE], sm);
    if (ops) atomicAdd(&ct[CT_SUM_OPS], ops);
    if (txt) atomicAdd(&ct[CT_SUM_TEXT], txt);
    if (cl) atomicAdd(&ct[CT_CELLS], cl);
  }
  uint32_t bad = 0;
  for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < n_all; i += gridDim.x * 1024u) {
    if (i + 3u < n_all) {
      const uint4 v = *reinterpret_cast<const uint4*>(status + i);
      bad += (v.x != WFA_ST_DONE) + (v.y != WFA_ST_DONE) + (v.z != WFA_ST_DONE) + (v.w != WFA_ST_DONE);
    } else {
      for (uint32_t j = i; j < n_all; ++j) bad += status[j] != WFA_ST_DONE;
    }
  }
  for (int d = 32; d > 0; d >>= 1) bad += __shfl_down(bad, d);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&ct[CT_UNFIN], (unsigned long long)bad);
}

// every stride-th entry of a pair list -> sample list
__global__ void k_sample(const uint32_t* __restrict__ list, uint32_t n_s, uint32_t stride, uint32_t* __restrict__ out) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n_s) out[gid] = list[(size_t)gid * stride];
}

// score per 1024 bases of the longer sequence (INT_MAX for pairs that did not finish)
__global__ void k_ratio(const uint32_t* __restrict__ list, uint32_t n, const uint32_t* __restrict__ status,
                        const int32_t* __restrict__ score, const WfaSeqPair* __restrict__ meta, int32_t* __restrict__ out) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const uint32_t pair = list[gid];
  const unsigned len = max(1u, max(meta[pair].pattern_len, meta[pair].text_len));
  out[gid] = (status[pair] == WFA_ST_DONE) ? (int32_t)(((long long)score[pair] * 1024 + len - 1) / len) : INT_MAX;
}

// per-pair budget = 1.02 * q/1024 * length + slack
__global__ void k_budget(const WfaSeqPair* __restrict__ meta, uint32_t n, int q, int slack, int margin_pct, int32_t* __restrict__ out) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const unsigned len = max(meta[gid].pattern_len, meta[gid].text_len);
  out[gid] = (int32_t)min(0x3FFFFFFFll, ((long long)q * len * margin_pct / 100) / 1024 + slack);
}

// pairs that are not DONE at the end of a call (must be none: every list is run to completion).  Grid-stride, four statuses per
// load: one thread per status took 41 us for 1M pairs -- a million threads for 4 MB.
__global__ void __launch_bounds__(256) k_count_unfinished(uint32_t n, const uint32_t* __restrict__ status, unsigned long long* __restrict__ count) {
  uint32_t bad = 0;
  for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < n; i += gridDim.x * 1024u) {
    if (i + 3u < n) {
      const uint4 v = *reinterpret_cast<const uint4*>(status + i);      // (the status array is 16-byte aligned, i a multiple of 4)
      bad += (v.x != WFA_ST_DONE) + (v.y != WFA_ST_DONE) + (v.z != WFA_ST_DONE) + (v.w != WFA_ST_DONE);
    } else {
      for (uint32_t j = i; j < n; ++j) bad += status[j] != WFA_ST_DONE;
    }
  }
  for (int d = 32; d > 0; d >>= 1) bad += __shfl_down(bad, d);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, (unsigned long long)bad);
}

// scores of a run with penalties divided by their common factor g -> scores under the caller's penalties
__global__ void k_scale_scores(int32_t* __restrict__ score, uint32_t n, int g) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n && score[gid] > 0) score[gid] *= g;
}

// moves the arena's bump pointer past a region a launch handed out statically (tier 5 with CIGARs), never beyond the arena
__global__ void k_bump(unsigned long long* top, unsigned long long units, unsigned long long cap) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t = *top + units; *top = t < cap ? t : cap; }
}

__global__ void k_iota(uint32_t* __restrict__ out, uint32_t first, uint32_t n) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n) out[gid] = first + gid;
}

__global__ void k_set_pending(const uint32_t* __restrict__ work, uint32_t n, uint32_t* __restrict__ status) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n) status[work ? work[gid] : gid] = WFA_ST_PENDING;
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Workgroup size of a list compaction.  Big lists: 1024 threads, one atomic per workgroup.  Lists of a batch of a
// pipelined call: one wavefront per workgroup, which finds a free wave slot next to the persistent wavefront kernel of
// the device's other lane (a 16-wave workgroup waits until a whole CU drains: measured 1.7 ms for a 5 us kernel).
inline uint32_t compact_block(uint32_t n) { return n <= (1u << 18) ? 64u : 1024u; }

}  // namespace

struct wfagpu_amd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 0;
  size_t lds_per_block_max = 0;
  size_t arena_cfg = 0, text_cfg = 0, arena_limit = 0, arena_limit_max = 0;
  wfagpu_amd_tuning_t tuning{};
  bool counters_zeroed = false;    // (the last call returned with its counters zeroed again behind its last read: the next one need not start with a memset)
  bool short_ascii_ok = true;      // (this call's ASCII buffer is below 16 GiB: tier 5 indexes it by 32-bit dwords)
  DevBuf packed, flags, status, cells, bt_final, list_a, list_b, list_c, list_d, list_e, work_ctr, sample, ratio, budget, counters, arena, ops, text_scratch, gring;
  // The results of a CIGAR call -- dense text, offsets, lengths -- alternate between two sets of buffers: the pointers a call
  // hands out stay valid through the NEXT call, so a pipelined caller copies the results of batch j to the host while the
  // kernels of batch j+1 run (launch_alignments*: the copy left the lanes' critical path).
  DevBuf text[2], cig_off[2], cig_len[2];
  DevBuf dbg;               // tuning.timed_barriers: barrier records of workgroup 0
  int dbg_waves = 0; unsigned dbg_records = 0;
  int out_set = 0;
  unsigned long long* h_counters = nullptr;  // pinned
  hipEvent_t ev_start = nullptr, ev_pack = nullptr, ev_a0 = nullptr, ev_a1 = nullptr, ev_b0 = nullptr, ev_b1 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr, ev_end = nullptr;
  wfagpu_amd_stats_t stats{};
  // budgets learnt from the sample of an earlier batch of the same stream (wfagpu_amd_hint_same_stream)
  bool same_stream = false;
  // (keyed by the length CLASS of the bucket -- the power of two above its longest pair: the budget is a score per 1024
  // bases, it carries over between batches whose longest reads differ by a few bases)
  struct SavedQ { unsigned bucket_hi; int q; int x, o, e, max_error; unsigned last_missed; } saved_q[8] = {};      // (last_missed: pairs of the last batch that exceeded these budgets)
  static unsigned length_class(unsigned len) { unsigned c = 1; while (c < len && c < (1u << 31)) c <<= 1; return c; }
  int n_saved_q = 0;
  uint32_t ct_used = 0;      // counters (bit = index) used since the call zeroed them all: zero_counter re-zeroes only those
  std::vector<Retired> retired;      // outgrown device buffers waiting for a moment when hipFree does not stall anybody
  size_t retired_bytes() const { size_t b = 0; for (const auto& r : retired) b += r.bytes; return b; }
  void free_retired() { for (auto& r : retired) hipFree(r.p); retired.clear(); }
  bool primed = false;
};

extern "C" {

int wfagpu_amd_create(wfagpu_amd_ctx_t** out, const wfagpu_amd_config_t* cfg) {
  if (!out) return -1;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  const int dev = cfg ? cfg->device : 0;
  if (dev < 0 || dev >= ndev) {
    fprintf(stderr, "[!] ERROR: HIP device %d not available (%d visible)\n", dev, ndev);
    return -1;
  }
  HIP_TRY(hipSetDevice(dev));
  wfagpu_amd_ctx* c = new wfagpu_amd_ctx();
  c->device = dev;
  // (a context that fails half-way is torn down again: wfagpu_amd_destroy copes with whatever exists so far)
  auto init = [&]() -> int {
    if (cfg && cfg->null_stream) {
      c->stream = nullptr;      // (the device's null stream: ordered with every blocking stream of the process)
    } else if (cfg && cfg->stream) {
      c->stream = static_cast<hipStream_t>(cfg->stream);
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      c->own_stream = true;
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    c->num_cus = prop.multiProcessorCount;
    c->lds_per_block_max = prop.sharedMemPerBlock;   // 160 KiB on gfx950 (checked, not assumed)
    c->arena_cfg = cfg ? cfg->arena_bytes : 0;
    c->text_cfg = cfg ? cfg->text_bytes : 0;
    c->arena_limit = cfg ? cfg->arena_limit_bytes : 0;
    c->arena_limit_max = cfg ? cfg->arena_limit_max_bytes : 0;
    if (cfg) c->tuning = cfg->tuning;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_counters), CT_PARTS_OFF + CT_PART_BYTES * CT_PARTS_PER_CU * (size_t)c->num_cus, hipHostMallocDefault));
    HIP_TRY(hipEventCreate(&c->ev_start)); HIP_TRY(hipEventCreate(&c->ev_pack));
    HIP_TRY(hipEventCreate(&c->ev_a0)); HIP_TRY(hipEventCreate(&c->ev_a1));
    HIP_TRY(hipEventCreate(&c->ev_b0)); HIP_TRY(hipEventCreate(&c->ev_b1));
    HIP_TRY(hipEventCreate(&c->ev_t0)); HIP_TRY(hipEventCreate(&c->ev_t1));
    HIP_TRY(hipEventCreate(&c->ev_end));
    // (the eight claim counters of the wavefront kernels, 64 bytes apart, sit behind the u64 slots: one memset zeroes both)
    if (c->counters.ensure(CT_PARTS_OFF + CT_PART_BYTES * CT_PARTS_PER_CU * (size_t)c->num_cus, c->stream)) return -1;
    return 0;
  };
  if (init()) { wfagpu_amd_destroy(c); return -1; }
  *out = c;
  return 0;
}

void wfagpu_amd_destroy(wfagpu_amd_ctx_t* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);      // (nullptr: the null stream)
  c->free_retired();
  for (DevBuf* b : {&c->packed, &c->flags, &c->status, &c->cells, &c->bt_final, &c->list_a, &c->list_b, &c->list_c, &c->list_d, &c->list_e, &c->work_ctr, &c->sample, &c->ratio, &c->budget,
                    &c->counters, &c->arena, &c->ops, &c->text[0], &c->text[1], &c->text_scratch, &c->cig_off[0], &c->cig_off[1], &c->cig_len[0], &c->cig_len[1], &c->gring, &c->dbg})
    b->release();
  if (c->h_counters) hipHostFree(c->h_counters);
  for (hipEvent_t ev : {c->ev_start, c->ev_pack, c->ev_a0, c->ev_a1, c->ev_b0, c->ev_b1, c->ev_t0, c->ev_t1, c->ev_end})
    if (ev) hipEventDestroy(ev);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

size_t wfagpu_amd_fill_packed_offsets(sequence_pair_t* metadata, size_t n) {
  size_t off = 0;
  for (size_t i = 0; i < n; ++i) {
    metadata[i].pattern_offset_packed = off;
    off += 4 * (((size_t)metadata[i].pattern_len + 15) / 16 + 1);
    metadata[i].text_offset_packed = off;
    off += 4 * (((size_t)metadata[i].text_len + 15) / 16 + 1);
  }
  return off;
}

int wfagpu_amd_pack_device(wfagpu_amd_ctx_t* c, const wfagpu_amd_batch_t* b, void* d_packed, unsigned char* d_flags) {
  if (!c || !b || !d_packed || !d_flags) return -1;
  HIP_TRY(hipSetDevice(c->device));
  wfa_launch_pack(b->d_sequences, reinterpret_cast<const WfaSeqPair*>(b->d_metadata), (uint32_t)b->num_pairs,
                  static_cast<uint32_t*>(d_packed), d_flags, c->stream, b->max_seq_len);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

void* wfagpu_amd_stream(const wfagpu_amd_ctx_t* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

void wfagpu_amd_debug_times(const wfagpu_amd_ctx_t* c, const void** d_records, unsigned int* records, int* waves) {
  if (d_records) *d_records = c ? c->dbg.p : nullptr;
  if (records) *records = (c && c->dbg.p) ? c->dbg_records : 0u;
  if (waves) *waves = c ? c->dbg_waves : 0;
}

void wfagpu_amd_trim(wfagpu_amd_ctx_t* c) {
  if (!c || c->retired.empty()) return;
  hipSetDevice(c->device);
  c->free_retired();
}

int wfagpu_amd_prime(wfagpu_amd_ctx_t* c) {
  if (!c) return -1;
  if (c->primed) return 0;
  HIP_TRY(hipSetDevice(c->device));
  // one empty launch per kernel translation unit: the runtime loads a code object at the first launch of any of its kernels
  wfa_prime_align(c->stream);
  wfa_prime_trace(c->stream);
  wfa_prime_pack(c->stream);
  wfa_prime_short(c->stream);
  LAUNCH_K(k_iota, dim3(1), dim3(64), 0, c->stream, static_cast<uint32_t*>(nullptr), 0u, 0u);      // (this file's)
  HIP_TRY(hipGetLastError());
  c->primed = true;
  return 0;
}

void wfagpu_amd_set_tuning(wfagpu_amd_ctx_t* c, const wfagpu_amd_tuning_t* tuning) {
  if (c) c->tuning = tuning ? *tuning : wfagpu_amd_tuning_t{};
}

void wfagpu_amd_hint_same_stream(wfagpu_amd_ctx_t* c, int on) {
  if (!c) return;
  c->same_stream = on != 0;
  if (!on) c->n_saved_q = 0;
}

void wfagpu_amd_last_stats(const wfagpu_amd_ctx_t* c, wfagpu_amd_stats_t* out) {
  if (c && out) *out = c->stats;
}

}  // extern "C"

namespace {

struct TierPlan { int tier; int width; int max_score; size_t lds; int blocks_per_cu; int wpe = 8; };

// Widest diagonal window an alignment of score <= S can need (see the kernel): (S - o)/e + 1,
// never more than every diagonal of the longest pair.
int window_width(long long S, int o, int e, unsigned max_seq_len) {
  const long long all = 2ll * max_seq_len + 1;
  if (S >= INT_MAX / 2) return (int)all;
  return (int)std::max<long long>(1, std::min<long long>(all, (S - o) / e + 1));
}

// Smallest tier whose LDS footprint fits a score budget S.
// few_pairs: the list is expected to hold fewer pairs than the device has workgroup slots (the re-run of a long-read batch's
// budget misses): what counts is the latency of ONE alignment, so wide wavefronts take sixteen waves instead of four.
bool plan_tier(const wfagpu_amd_ctx* c, WfaAlignParams& p, int max_score, unsigned max_seq_len, bool bt, bool raw, TierPlan* out, bool few_pairs = false,
               uint32_t n_pairs = 0) {
  p.max_score = max_score;
  if (p.band_width > 0) {
    // Adaptive band: a ring row holds the band_width diagonals of its score, stored relative to the row's own lower limit,
    // between two guard zones of 4 dm + 2 NULL cells (see the kernel: the band moves by at most two diagonals per score),
    // plus the padding chunk of the lean cells.
    p.rs = ((p.band_width + 2 * (4 * p.dm + 2) + 2 + 1) & ~1) + WFA_RING_ROW_PAD;
    // (the banded kernels keep their row book in one register per wave, lane = score & 63: ring depths to 64)
    if (max_seq_len > 32766u || max_score > 30000 || p.dm > 64) return false;
    // Waves per alignment (1, 4 or 16) by what the DEVICE ends up holding, not by the band width alone (round 5).  A banded row
    // is band_width / 64 chunks whatever the score, every chunk a chain of dependent LDS round trips, so the kernel lives on
    // the waves a SIMD can switch between: 16k x 10 kbp pairs at a band of 512 on one wave each are 10 rings = 10 waves per CU
    // (LDS-bound, 2.5 per SIMD: 355 G cells/s against 607 G for the exact four-wave tier on the same pairs); 1024 x 30 kbp
    // pairs one wave each are ONE wave per SIMD.  The smallest tier that keeps 16 waves per CU resident (four per SIMD) -- with
    // the workgroups LDS allows and the pairs the launch has --, else the one that keeps the most.
    const int min_t = (c->tuning.min_tier >= 1 && c->tuning.min_tier <= 2) ? c->tuning.min_tier : 0;      // (test hook)
    const uint32_t pairs_per_cu = n_pairs ? std::max<uint32_t>(1u, (n_pairs + (uint32_t)c->num_cus - 1u) / (uint32_t)c->num_cus) : 1u << 20;
    int best_t = -1, best_waves = -1, best_nb = 0; size_t best_lds = 0;
    // (candidates by wavefronts per alignment: 1, 2, 4, 16 = tiers 0, 6, 1, 2; tuning.band_tier 1..4 forces one of them: A/B hook)
    static const int cand_tier[4] = {0, 6, 1, 2}, cand_nw[4] = {1, 2, 4, 16};
    const int forced_i = (c->tuning.band_tier >= 1 && c->tuning.band_tier <= 4) ? c->tuning.band_tier - 1 : -1;
    const int first_i = forced_i >= 0 ? forced_i : (min_t == 0 ? 0 : (min_t == 1 ? 2 : 3));
    for (int i = first_i; i < (forced_i >= 0 ? forced_i + 1 : 4); ++i) {
      const int t = cand_tier[i], nw = cand_nw[i];
      if (forced_i < 0 && i > first_i && p.band_width < 48 * nw) break;            // (a wave without a chunk of its own only waits at the barrier)
      const size_t lds = wfa_align_lds_bytes(p, t);
      if (lds > c->lds_per_block_max) continue;
      const int nb = wfa_align_max_blocks_per_cu(t, bt, false, true, lds);
      if (nb < 1) continue;
      const int waves = (int)std::min<uint32_t>(32u, std::min<uint32_t>((uint32_t)nb, pairs_per_cu) * (uint32_t)nw);
      if (waves > best_waves) { best_t = t; best_waves = waves; best_nb = nb; best_lds = lds; }
      if (waves >= 16) break;
    }
    if (best_t < 0) return false;
    *out = {best_t, p.band_width, max_score, best_lds, best_nb};
    return true;
  }
  const int width = window_width(max_score, p.oe - p.e, p.e, max_seq_len);
  // Short wavefronts, score only (tier 5, short_kernel.hip): when the diagonal window of the budget fits 16 or 32 lanes, four or two
  // alignments share a wavefront and the rings live in registers (BASELINE configs[1]: 150 bp reads, budgets of ~14 once
  // they are tuned).  Pairs whose own window is wider (large |tlen - plen|) come back as BAND failures and go on as always.
  if (!raw && !(p.ascii && !c->short_ascii_ok) && c->tuning.min_tier == 0 && !(bt && c->tuning.no_short_cigar) && wfa_short_supported(p.x, p.oe, p.e) && width + 1 <= 32 && max_score <= 30000) {
    const int lanes = width + 1 <= 16 ? 16 : 32;
    p.rs = 0;
    const size_t lds = wfa_short_lds_bytes(p, lanes, bt);
    if (lds <= 40u << 10) {
      // (what really fits a CU: the launch below cuts its grid so that every wavefront is resident from the start and all of
      // them run the same number of iterations)
      const int nb = std::min(32, wfa_short_max_blocks_per_cu(p, lanes, bt));
      if (nb >= 1) {
        *out = {5, width, max_score, lds, nb, lanes};      // (wpe carries the lanes per alignment)
        return true;
      }
    }
  }
  // dm guard cells on each side of a row (see the kernel's lean path); the 16-bit LDS tiers add a chunk of padding
  const int rs_plain = (width + 1 + 2 * p.dm + 1) & ~1;
  p.rs = rs_plain + WFA_RING_ROW_PAD;
  const bool i16_ok = max_seq_len <= 32766u && max_score <= 30000;
  const size_t budget[3] = {40u << 10, 80u << 10, c->lds_per_block_max};
  // tuning.min_tier (tests): skip the smaller tiers so that the rarely needed ones get exercised
  const int min_tier = c->tuning.min_tier;
  for (int t = min_tier; t < 3 && i16_ok; ++t) {
    const size_t lds = wfa_align_lds_bytes(p, t);
    if (lds > budget[t]) continue;
    // a single wavefront sweeps up to ~16 chunks per score before more waves pay off
    if (t == 0 && (width > 1024 || p.dm > 64)) continue;
    if (t == 1 && width > 8192) continue;
    if (t == 1 && few_pairs && width >= 1024 && !min_tier && wfa_align_lds_bytes(p, 2) <= budget[2]) continue;
    int nb = wfa_align_max_blocks_per_cu(t, bt, raw, false, lds);
    if (nb < 1) continue;
    // Occupancy-matched instantiation of the one-wave kernels: LDS decides how many rings a CU holds; compiled for 8 waves
    // per SIMD the kernel lives on 64 VGPRs and 78 SGPRs (123 scalar spills + scratch), which only pays when 8 waves per
    // SIMD really are resident.  The smallest of 4 / 6 / 7 / 8 that keeps all the rings LDS allows but at most one.
    int wpe = 8;
    if (t == 0 && !raw) {
      const int forced = c->tuning.waves_per_simd;
      for (int w : {4, 6, 7, 8}) if (4 * w >= nb - 1) { wpe = w; break; }
      // (gap extensions > 1: the general lean loop carries more scalar state than the e == 1 loop; compiled for 8 waves per SIMD -- 64
      // vector registers, the scalar spills go through them -- it runs slower with 31-32 rings per CU than the 7-wave build does with 28:
      // 1M x 1 kbp under (3,1,4) 27.85 against 25.3 ms, 500k x 400 bp under (5,3,2) 7.0 against 6.5 ms: scratch/wpe_probe.py)
      if (p.e != 1 && wpe == 8) wpe = 7;
      if (forced == 4 || forced == 6 || forced == 7 || forced == 8) wpe = forced;
      if (wpe != 8) nb = std::min(nb, wfa_align_max_blocks_per_cu(t, bt, raw, false, lds, wpe));
      if (nb < 1) continue;
    }
    // One wave per alignment needs its own ring: when LDS leaves fewer than 2.5 wavefronts per SIMD and the rows are wide
    // enough to share, four waves per alignment (same LDS per workgroup, 4x the resident waves) win although every wave
    // repeats the per-score bookkeeping: 16k x 10 kbp @ 3 % (24 KB rings, 6 per CU): 25.6 -> 20.3 ms (final kernels: 16.4);
    // 5 kbp @ 4 % (9 rings per CU): 31.2 -> 29.9 ms with four waves; 3 kbp @ 5 % (12 rings): one wave, 16.1 against 19.7 ms
    // (profiles/r02/mid_lengths.txt).  tuning.t0_min_blocks: A/B.
    const int t0_min_blocks = c->tuning.t0_min_blocks > 0 ? c->tuning.t0_min_blocks : 10;
    if (t == 0 && !min_tier && nb < t0_min_blocks && width >= 384) continue;
    if (t == 1 && !raw && c->tuning.exact_two_waves) {
      // A/B hook: TWO waves per alignment where four would run (same ring, same LDS; half the per-score bookkeeping per chunk, half the waves)
      const size_t lds2 = wfa_align_lds_bytes(p, 6);
      const int nb2 = wfa_align_max_blocks_per_cu(6, bt, raw, false, lds2);
      if (nb2 >= 1) { *out = {6, width, max_score, lds2, nb2, wpe}; return true; }
    }
    *out = {t, width, max_score, lds, nb, wpe};
    return true;
  }
  // hybrid ring: everything but the D rows in LDS (one workgroup of 16 waves per CU)
  if (i16_ok && !raw && min_tier != 3) {
    const size_t lds_h = wfa_align_lds_bytes(p, 4);
    if (lds_h <= c->lds_per_block_max) {
      const int nb = wfa_align_max_blocks_per_cu(4, bt, raw, false, lds_h);
      if (nb >= 1) { *out = {4, width, max_score, lds_h, nb}; return true; }
    }
  }
  p.rs = rs_plain;
  const size_t lds = wfa_align_lds_bytes(p, 3);
  if (lds > c->lds_per_block_max) return false;   // sequences themselves do not fit LDS
  const int nb = wfa_align_max_blocks_per_cu(3, bt, raw, false, lds);
  *out = {3, width, max_score, lds, std::max(1, std::min(nb, 2))};
  return true;
}

// (n_parts > 0: with the partial sums of a tier-5 launch of n_parts wavefronts -- one copy)
int read_counters(wfagpu_amd_ctx* c, uint32_t n_parts = 0) {
  const size_t bytes = n_parts ? CT_PARTS_OFF + CT_PART_BYTES * n_parts : CT_N * sizeof(unsigned long long);
  HIP_TRY(hipMemcpyAsync(c->h_counters, c->counters.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// Called before every use of a counter.  All counters are zeroed together when a call starts; a counter is zeroed again here
// only when it has been used since (every memset is a kernel launch of its own: a simple call -- one pass, one chain -- now
// has none of them).
int zero_counter(wfagpu_amd_ctx* c, int idx, int count = 1) {
  const uint32_t mask = ((1u << count) - 1u) << idx;
  if (c->ct_used & mask) HIP_TRY(hipMemsetAsync(static_cast<unsigned long long*>(c->counters.p) + idx, 0, sizeof(unsigned long long) * count, c->stream));
  c->ct_used |= mask;
  return 0;
}

}  // namespace

extern "C" int wfagpu_amd_align_device(wfagpu_amd_ctx_t* c, const wfagpu_amd_batch_t* b,
                                       affine_penalties_t pen, int max_error, int band, int band_width,
                                       bool compute_cigar, int32_t* d_scores,
                                       const char** d_text, const unsigned long long** d_off,
                                       const unsigned int** d_len) {
  if (!c || !b || !d_scores) return -1;
  if (pen.x <= 0 || pen.o < 0 || pen.e <= 0) {
    fprintf(stderr, "[!] ERROR: penalties must be x>0, o>=0, e>0 (got %d,%d,%d)\n", pen.x, pen.o, pen.e);
    return -1;
  }
  const bool want_band = band > 0 && band_width > 0;
  // Penalties with a common factor g (WFA2's default 4,6,2) describe the same alignments as the penalties divided by g,
  // with every score multiplied by g: same optimal paths, same tie-breaks, hence the same CIGARs.  Running the reduced
  // set avoids the (g-1)/g of all scores that have no wavefront at all and halves the score-indexed work.
  int pen_scale = 1;
  {
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    const int g = gcd(gcd(pen.x, pen.o), pen.e);
    if (g > 1) {
      pen_scale = g; pen.x /= g; pen.o /= g; pen.e /= g; max_error = std::max(1, max_error / g);
      if (want_band) band = std::max(1, band / g);     // (the re-centring period counts scores)
    }
  }
  HIP_TRY(hipSetDevice(c->device));
  // (buffers outgrown during this call are retired, not freed: see DevBuf)
  if (c->retired_bytes() > ((size_t)1 << 30)) c->free_retired();
  struct RetireScope { RetireScope(std::vector<Retired>* r) { t_retire = r; } ~RetireScope() { t_retire = nullptr; } } retire_scope(&c->retired);
  const uint32_t n = (uint32_t)b->num_pairs;
  c->stats = wfagpu_amd_stats_t{};
  if (d_text) *d_text = nullptr;
  if (d_off) *d_off = nullptr;
  if (d_len) *d_len = nullptr;
  if (n == 0) return 0;
  if (max_error < 1) max_error = 1;
  hipStream_t st = c->stream;
  unsigned long long* ct = static_cast<unsigned long long*>(c->counters.p);

  // ---- buffers ------------------------------------------------------------
  // (a batch that arrives packed -- wfagpu_amd_batch_t::d_packed, all ACGT by contract -- needs neither the words nor the flags)
  const bool prepacked = b->d_packed != nullptr;
  if (!prepacked) {
    if (c->packed.ensure(b->packed_bytes + 16, st)) return -1;
    if (c->flags.ensure((size_t)2 * n, st)) return -1;
  }
  if (c->status.ensure((size_t)4 * n, st)) return -1;
  if (c->cells.ensure((size_t)4 * n, st)) return -1;
  if (c->list_a.ensure((size_t)4 * n, st)) return -1;
  if (c->list_b.ensure((size_t)4 * n, st)) return -1;
  if (c->list_c.ensure((size_t)4 * n, st)) return -1;
  if (c->list_d.ensure((size_t)4 * n, st)) return -1;
  if (compute_cigar) {
    if (c->bt_final.ensure((size_t)4 * n, st)) return -1;
    if (c->cig_off[c->out_set].ensure((size_t)8 * n, st)) return -1;
    if (c->cig_len[c->out_set].ensure((size_t)4 * n, st)) return -1;
    // (the OTHER set of output buffers, the next call's: made now when it does not exist yet -- it holds nobody's results --, so that
    // the second call of a context is not the one that pays for three more allocations: 4.8 ms against a median of 2.7 for a 100k-pair call)
    if (!c->cig_off[c->out_set ^ 1].p && c->cig_off[c->out_set ^ 1].ensure((size_t)8 * n, st)) return -1;
    if (!c->cig_len[c->out_set ^ 1].p && c->cig_len[c->out_set ^ 1].ensure((size_t)4 * n, st)) return -1;
  }
  const unsigned batch_max_len = std::max(1u, b->max_seq_len);
  unsigned max_len = batch_max_len;    // of the pairs being run: the whole batch, or one length bucket of it
  const int oe = pen.o + pen.e;
  // widest RLE item of a CIGAR of this batch: the digits of the longest possible run + the operation
  int item_chars = 2;
  for (unsigned v = batch_max_len; v >= 10; v /= 10) ++item_chars;
  item_chars = std::max(item_chars, 6);      // (the emit kernels' one-word fast path writes 4 bytes per item)

  WfaAlignParams ap{};
  ap.packed = prepacked ? static_cast<const uint32_t*>(b->d_packed) : static_cast<const uint32_t*>(c->packed.p);
  ap.meta = reinterpret_cast<const WfaSeqPair*>(b->d_metadata);
  ap.x = pen.x; ap.oe = oe; ap.e = pen.e;
  ap.dm = std::max(pen.x, oe) + 1;
  ap.de = pen.e + 1;
  { int bk = 64; while (bk < ap.dm) bk <<= 1; ap.book_mask = bk - 1; }
  ap.seq_words_cap = (int)((max_len + 15) / 16 + 1);
  ap.score = d_scores;
  ap.status = static_cast<uint32_t*>(c->status.p);
  ap.cells = static_cast<uint32_t*>(c->cells.p);
  ap.work_counter = reinterpret_cast<unsigned int*>(static_cast<char*>(c->counters.p) + CT_BYTES);
  ap.arena_top = ct + CT_ARENA;
  ap.launch_cells = ct + CT_LCELLS;
  ap.no_lean = c->tuning.careful_only ? 1 : 0;
  ap.chunk_units = 256;   // 4 KiB refills
  ap.bt_final_row = static_cast<uint32_t*>(c->bt_final.p);

  // Expected backtrace bytes per pair: a quarter of the score-budget square of origin bytes plus row
  // headers; refined from what finished pairs really used after every pass.  Passes launch only as
  // many pairs as the arena is expected to hold, so a small arena costs passes, not repeated work.
  double est_pair_bytes = 0.25 * ((double)std::min<unsigned>(max_error, 2 * max_len) + 1) * ((double)std::min<unsigned>(max_error, 2 * max_len) + 1) +
                          16.0 * std::min<unsigned>(max_error, 2 * max_len) + 4096.0;
  // The one-wave exact tier hands every alignment a block of TILES up front (align_kernel.hip, TILED): a 64-byte tile per 4 scores x 16
  // diagonals over the whole window of the budget, whatever part of it the wavefronts visit -- about twice the diamond's bytes.
  auto tiled_pair_bytes = [&](const long long B) -> double {
    const int w = window_width(B, pen.o, pen.e, batch_max_len);
    if (w > 1024) return 0.0;      // (wider wavefronts run on the multi-wave tiers: rows behind a row table)
    return 64.0 * (double)((w + 15) / 16) * (double)(B / 4 + 1) + 512.0;
  };
  est_pair_bytes = std::max(est_pair_bytes, tiled_pair_bytes(std::min<long long>(max_error, 2ll * batch_max_len * std::max(pen.x, pen.e))));
  if (compute_cigar) {
    // arena: expected need, bounded by configuration / free memory / 32-bit unit addressing
    size_t want = c->arena_cfg;
    if (!want) {
      want = (size_t)(est_pair_bytes * 1.1 * n);
      if (want > c->arena.cap) {
        // (the driver is only asked when the arena has to grow: the query costs as much as a kernel launch or two)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        want = (size_t)std::min<double>((double)want, 0.45 * (double)(free_b + c->arena.cap));
      }
      want = std::max<size_t>(want, (size_t)64 << 20);
      if (c->arena_limit) {
        // (the starting cap keeps first calls cheap; it gives way, up to the maximum, when it could not even hold the
        // pairs that are in flight at once -- long reads: 1024 x 30 kbp @ 10 % went through five failed passes and
        // re-allocations, ~2 s, on a process's first call)
        const double inflight = std::min<double>(n, 2.0 * c->num_cus) * est_pair_bytes * 2.0;
        const size_t lim = std::max<size_t>(c->arena_limit, (size_t)std::min<double>((double)c->arena_limit_max, inflight));
        want = std::min(want, std::max<size_t>(lim, (size_t)64 << 20));
      }
    }
    want = std::min<size_t>(want, ((size_t)1 << 36) - 4096);
    if (c->arena.cap < want && c->arena.ensure(want, st)) return -1;
    ap.arena = static_cast<uint8_t*>(c->arena.p);
    ap.arena_units = c->arena.cap / 16 - 8;      // (128 bytes of slack: the walk reads row-table entries a line at a time)
  }

  // Long enough reads are packed by the wavefront kernels themselves while they stage them (WfaAlignParams::ascii): no pack
  // kernel in front of the first wavefront launch (0.63 ms per 1M x 1 kbp pairs).  Short reads too where tier 5 (several
  // alignments per wavefront, short_kernel.hip: the sixteen bytes of a word requested ahead instead of the word) takes the penalties:
  // 18 us + the gap between two launches of a 168 us step of 100k configs[1] pairs.  (ASCII dword indices are 32-bit there.)
  // (sequences_bytes may be missing from a caller's batch record: four times the packed words bound the ASCII layout from above)
  c->short_ascii_ok = std::max<size_t>(b->sequences_bytes, b->packed_bytes * 4) < ((size_t)1 << 34);
  const bool short_fuses = wfa_short_supported(pen.x, oe, pen.e) && !c->tuning.min_tier && !(compute_cigar && c->tuning.no_short_cigar) && c->short_ascii_ok;
  const bool fused_pack = !prepacked && (b->max_seq_len >= 512u || short_fuses) && !c->tuning.no_fused_pack;
  // Every status starts as PENDING = 0: written by the pack kernel where it runs (it visits every pair anyway), by a memset (a
  // launch of its own, ~5 us + a gap) where it does not -- queued by whoever reads the array first (ensure_status), and not at all
  // when the call's first launch is tier 5 over the whole batch: that kernel takes every pair as PENDING and leaves every pair with
  // a status.
  bool status_unset = prepacked || fused_pack;
  auto ensure_status = [&]() -> int {
    if (status_unset) { HIP_TRY(hipMemsetAsync(c->status.p, 0, (size_t)4 * n, st)); status_unset = false; }
    return 0;
  };
  // Inherited budgets: the array is filled (k_budget) by the first launch that needs it; tier 5 applies the rule itself.
  struct { bool unset; int q, slack, margin; } budget_rule{false, 0, 0, 0};
  unsigned long long cells_host = 0;      // cells of launches whose partial sums were added up on the host (run_list: a chain of one tier-5 launch)
  hipEvent_t call_end = c->ev_end;        // the event that carries the end stamp of the call's last kernel
  bool ct_clean = true;                   // no kernel of this call has written one of the device's counters yet (they are all zero)
  // (all counters start a call as zero.  A call that ends well zeroes them behind its last read -- queued when the caller already has
  // its results, executed while the host prepares the next call -- instead of in front of the next call's first kernel: a launch of
  // ~3 us and the gap behind it, of a 150 us step of 100k short reads)
  if (!c->counters_zeroed) HIP_TRY(hipMemsetAsync(c->counters.p, 0, CT_BYTES + 8 * 64, st));
  else if (c->tuning.verify_counters) {
    // (test hook: the call before this one said that it left the counters -- the u64 slots and the eight claim counters -- zeroed)
    HIP_TRY(hipMemcpyAsync(c->h_counters, c->counters.p, CT_BYTES + 8 * 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const unsigned char* hb = reinterpret_cast<const unsigned char*>(c->h_counters);
    for (size_t i = 0; i < CT_BYTES + 8 * 64; ++i)
      if (hb[i]) { fprintf(stderr, "[!] ERROR: the device's counter block was left dirty by the call before (byte %zu)\n", i); return -1; }
  }
  c->counters_zeroed = false;
  c->ct_used = 0;
  // (ev_start -- the start of the call on the device -- is the start stamp of the FIRST kernel of the call, carried by its own dispatch
  // packet: the pack kernel, or the first wavefront launch.  A hipEventRecord is a barrier packet of its own: ~5 us in front of the
  // first kernel of every call)
  bool have_start = false;
  ap.ascii = fused_pack ? b->d_sequences : nullptr;
  ap.n_raw = ct + CT_NRAW;
  if (!prepacked && !fused_pack) {
    // (ev_pack: the end of the pack kernel, stamped by its own dispatch packet -- wfa_launch_timed, wfa_device.h)
    have_start = true;
    wfa_launch_pack(b->d_sequences, ap.meta, n, static_cast<uint32_t*>(c->packed.p), static_cast<uint8_t*>(c->flags.p), st, b->max_seq_len, c->ev_start, c->ev_pack,
                    static_cast<uint32_t*>(c->status.p), ct + CT_NRAW);      // (flagged pairs: status ALPHABET, counted)
  }

  float align_ms = 0.f, trace_ms = 0.f;
  uint32_t grid_cap = UINT32_MAX;   // lowered when a pass makes no progress for lack of arena
  bool cigar_now = compute_cigar;   // (false while the sample of the auto-budget step runs: scores only)
  unsigned long long arena_units_call = 0;
  unsigned long long text_used = 0;
  unsigned long long sample_cells_call = 0;    // cells of auto-budget samples whose pairs were aligned again
  long long unfinished_at_sync = -1;           // pairs not DONE as of the last chain's synchronisation; -1: work was queued since
  int rc = 0;

  // Runs one list of pairs to completion: passes bounded by the arena, tier escalation inside a pass,
  // backtrace + CIGAR text for what finished.  `budgets` (optional, per pair) is tried first.
  // `alt0`/`alt1`: the two lists the passes ping-pong between for what is left over (NOMEM pairs + pairs not launched
  // yet).  The caller picks them so that neither is a list it still needs (the sampled run of the auto-budget step
  // must not touch the bucket's own list).
  // `pending` == nullptr: the identity list 0..n_pending-1, UNFILTERED -- it may hold pairs flagged for the byte-compare
  // class; the first launch over such a list skips whatever is not PENDING.
  //
  // A pass is a sequence of CHAINS, each enqueued without a host round trip: wavefront kernel -> compaction of its
  // failures [-> the re-run of the pairs that missed their auto-tuned budget, its length read on the device ->
  // compaction] -> backtrace + CIGAR text of what finished -> compaction of the pairs that ran out of arena -> ONE
  // synchronisation.  Further chains (wider tiers) follow only for pairs that are still unfinished then.  (Six round
  // trips per batch before: with two lanes per device the lanes fell into lockstep, both waiting on the host at once.)
  auto run_list = [&](uint32_t* pending, uint32_t n_pending, const bool raw, const int32_t* budgets, const int budget_cap,
                      uint32_t* alt0, uint32_t* alt1, const bool allow_band = true, const bool speculate = true) -> int {
  grid_cap = UINT32_MAX;
  const bool unfiltered = pending == nullptr;
  if (budgets) {
    // tight budgets make the wavefront a diamond: at most half the budget square of origin bytes
    const double B = budget_cap;
    est_pair_bytes = std::max(0.5 * (B + 1) * (B + 1) + 16.0 * B + 2048.0, tiled_pair_bytes(budget_cap));
  }
  while (n_pending > 0) {
    c->stats.sub_batches++;
    // pairs launched in this pass: what the arena is expected to hold (at least one wave of workgroups)
    uint32_t n_pass = n_pending;
    if (cigar_now) {
      const double fit = (double)c->arena.cap / std::max(1.0, est_pair_bytes);
      n_pass = (uint32_t)std::min<double>(n_pending, std::max<double>(fit, 4.0 * c->num_cus));
    }
    if (zero_counter(c, CT_ARENA)) return -1;
    if (zero_counter(c, CT_NOMEM)) return -1;
    uint32_t* nxt_pending = (pending == alt0) ? alt1 : alt0;
    uint32_t* cur = pending;
    uint32_t n_cur = n_pass;
    uint32_t* spare[2] = {static_cast<uint32_t*>(c->list_a.p), static_cast<uint32_t*>(c->list_b.p)};
    int flip = 0;
    int max_score = budgets ? std::min(budget_cap, max_error) : max_error;
    bool budget_round = budgets != nullptr;
    int round = 0;
    while (n_cur > 0) {
      // ---- one chain ------------------------------------------------------------------------------------------------
      unfinished_at_sync = -1;      // (work is being queued: the last count is history)
      uint32_t* const chain_list = cur;
      const uint32_t n_chain = n_cur;
      struct Link { int tier; int ct_list; int ct_cells; hipEvent_t e0, e1; bool banded; bool budgeted; bool first_round; uint32_t n_in; bool walked; } link[2];
      int n_links = 0;
      bool fused_tail = false; uint32_t* tail_out = nullptr; unsigned long long* tail_count = nullptr; unsigned long long* tail_parts_out = nullptr;
      bool solo = false, parts_on_host = false, skip_read = false; uint32_t parts_n = 0;      // (see the launch below)
      bool slots_fit = false;      // (a tier-5 launch with CIGARs whose static arena slots all lie inside the arena: no pair can run out of arena)
      const bool ct_clean_before = ct_clean;      // (no kernel of this call has written a counter so far)
      ct_clean = false;
      long long s_hi = 0;                       // no pair of this chain finishes with a larger score
      const unsigned long long* cur_len_dev = nullptr;
      for (;;) {
        Link& L = link[n_links];
        L = {0, n_links ? CT_LIST2 : CT_LIST, n_links ? CT_LCELLS2 : CT_LCELLS, n_links ? c->ev_b0 : c->ev_a0, n_links ? c->ev_b1 : c->ev_a1,
             false, budget_round, round == 0, n_cur, false};
        if (!have_start) { L.e0 = c->ev_start; have_start = true; }      // (the call's first kernel: its start stamp is the call's)
        TierPlan tp;
        ap.budget = budget_round ? budgets : nullptr;
        // banded first attempt (packed class only); whatever it cannot finish goes to the exact tiers
        // (with tuned budgets the band keeps them: the reach interval of a tight budget makes the banded wavefront shrink
        // towards the end like the exact one; a pair whose banded score exceeds its budget is re-run exactly like any miss)
        ap.band_width = (want_band && allow_band && !raw && round == 0) ? band_width : 0;
        ap.band_period = band;
        // (a banded launch runs under the caller's max_error, not under the tuned per-pair budgets: a budget narrows nothing
        // there -- every score has band_width diagonals whatever it is, the reference's windows are not clipped by a budget's
        // reach -- and a pair whose banded score passed its budget went through a second, exact alignment for nothing:
        // BASELINE configs[3] with the band forced, 198 of 16 384 pairs re-run on sixteen waves each, 2.5 of a 20 ms step)
        if (ap.band_width > 0 && !plan_tier(c, ap, std::min(max_error, 30000), max_len, cigar_now, raw, &tp, false, n_cur)) ap.band_width = 0;
        L.banded = ap.band_width > 0;
        if (L.banded) { ap.budget = nullptr; L.budgeted = false; }
        // (the speculative re-run of budget misses: a percent or two of the chain's pairs)
        const bool few_pairs = n_links == 1 && link[0].budgeted && (n_chain / 50u) <= 2u * (unsigned)c->num_cus;
        if (ap.band_width == 0 && !plan_tier(c, ap, max_score, max_len, cigar_now, raw, &tp, few_pairs)) {
          fprintf(stderr, "[!] ERROR: sequences of %u bases do not fit the LDS staging area\n", max_len);
          return -1;
        }
        // The last resort (tier 3: the whole ring in HBM / L2) costs 2.4x the hybrid tier per cell (1024 x 30 kbp pairs whose scores are
        // ~8200: 170 ms under -e 12000 against 70 ms under -e 9000, BENCH_r05 ont_banded.long_read_shaped) -- and the budget that sent
        // the pairs there is the caller's guess of a CEILING (the CLI's default: a tenth of the length times the largest penalty).
        // Where no budgets were tuned (batches too small for a sample, the exact re-run of a banded launch's misses) and the caller's
        // budget only fits tier 3, the chain first runs under the LARGEST budget an LDS tier holds; whoever misses it is escalated like
        // any other failure (4x the budget: the HBM ring then).  Results stay exact: a pair that finishes within a budget has its
        // optimal score, whatever the budget was (the window argument of the kernel).
        if (ap.band_width == 0 && tp.tier == 3 && !budget_round && !c->tuning.min_tier && !raw && max_score > 256 && max_score <= 30000 && max_len <= 32766u) {
          int lo_s = 128, hi_s = max_score;      // fits(lo_s) assumed, fits(hi_s) false
          TierPlan tq;
          auto fits = [&](const int S) { WfaAlignParams probe = ap; return plan_tier(c, probe, S, max_len, cigar_now, raw, &tq, few_pairs) && tq.tier != 3; };
          if (fits(lo_s)) {
            while (hi_s - lo_s > 8) { const int mid = lo_s + (hi_s - lo_s) / 2; if (fits(mid)) lo_s = mid; else hi_s = mid; }
            // (worth a pass of its own only when it covers most of the caller's range: below half of it the pairs that the caller
            // expects near its ceiling would all run twice)
            if (lo_s >= max_score / 2 && plan_tier(c, ap, lo_s, max_len, cigar_now, raw, &tp, few_pairs) && tp.tier != 3) max_score = lo_s;
            else if (!plan_tier(c, ap, max_score, max_len, cigar_now, raw, &tp, few_pairs)) return -1;
          }
        }
        L.tier = tp.tier;
        if (tp.tier == 3 || tp.tier == 4) {
          ap.ring16 = max_len <= 32766u ? 1 : 0;
          const size_t stride = tp.tier == 4 ? ((((size_t)ap.de * ap.rs * 2) + 255) & ~(size_t)255)     // hybrid: the D ring only
                                             : ((((size_t)(ap.dm + 2 * ap.de) * ap.rs * (ap.ring16 ? 2 : 4)) + 255) & ~(size_t)255);
          const int g = (int)std::min<uint32_t>(n_cur, (uint32_t)(c->num_cus * tp.blocks_per_cu));
          if (c->gring.ensure(stride * g, st)) return -1;
          ap.gring = c->gring.p; ap.gring_stride = stride;
        }
        // (tuning.kernel_walk -- the one-wave kernels walking their own alignments back, measured a loss in round 5 -- went with the row
        // table it read: round 6's tiles make wfa_walk_kernel itself cheap)
        L.walked = false;
        ap.walk_in_kernel = 0;
        ap.cigar_off = static_cast<unsigned long long*>(c->cig_off[c->out_set].p);
        ap.cigar_len = static_cast<uint32_t*>(c->cig_len[c->out_set].p);
        ap.dbg_times = nullptr; ap.dbg_cap = 0;
        if (c->tuning.timed_barriers && round == 0 && cigar_now && !raw && !L.banded && (tp.tier == 1 || tp.tier == 4)) {
          const int nw = tp.tier == 1 ? 4 : 16;
          const uint32_t cap = 1u << 16;
          if (c->dbg.ensure((size_t)cap * nw * 24, st)) return -1;
          HIP_TRY(hipMemsetAsync(c->dbg.p, 0, (size_t)cap * nw * 24, st));
          c->dbg_waves = nw; c->dbg_records = cap;
          ap.dbg_times = static_cast<unsigned long long*>(c->dbg.p); ap.dbg_cap = cap;
        }
        if (round == 0) { c->stats.lds_bytes_tier0 = tp.lds; c->stats.blocks_per_cu_tier0 = tp.blocks_per_cu; c->stats.waves_per_simd_tier0 = tp.wpe; }
        ap.budget_q = 0;
        if (ap.budget && budget_rule.unset) {
          if (tp.tier == 5) { ap.budget = nullptr; ap.budget_q = budget_rule.q; ap.budget_slack = budget_rule.slack; ap.budget_margin = budget_rule.margin; }
          else {
            LAUNCH_K(k_budget, dim3(cdiv(n, 256)), dim3(256), 0, st, ap.meta, n, budget_rule.q, budget_rule.slack, budget_rule.margin, static_cast<int32_t*>(c->budget.p));
            budget_rule.unset = false;
          }
        }
        ap.work = cur; ap.n_work = n_cur; ap.n_work_dev = cur_len_dev;
        ap.only_pending = (unfiltered && round == 0) ? 1 : 0;
        if (status_unset) {
          // (the call's first launch)
          if (tp.tier == 5 && unfiltered && round == 0 && n_cur == n) { ap.only_pending = 0; status_unset = false; }
          else if (ensure_status()) return -1;
        }
        ap.launch_cells = ct + L.ct_cells;
        const int bpc_cap = c->tuning.max_blocks_per_cu > 0 ? c->tuning.max_blocks_per_cu : 1 << 20;     // (occupancy experiments)
        // (tier 5: 64 / lanes alignments per wavefront)
        const uint32_t units = tp.tier == 5 ? cdiv(n_cur, 64u / (uint32_t)tp.wpe) : n_cur;
        int grid = (int)std::min<uint32_t>(std::min<uint32_t>(units, grid_cap), (uint32_t)(c->num_cus * std::min(tp.blocks_per_cu, bpc_cap)));
        // (tier 5 hands its work out by a grid-stride loop over items of equal cost: a grid that is not resident as a whole, or whose
        // wavefronts do not all run the same number of iterations, ends in a tail -- 100k configs[1] pairs = 25 000 items on 8192
        // wavefronts of which 7168 were resident: six rounds of iterations for 3.5 rounds of work.  So: as many wavefronts as are
        // resident, the iterations evened out.  Through round 4 a wavefront also ran at least seven iterations -- fewer, longer-lived
        // wavefronts measured faster: what they saved was the atomic every wavefront ended with on the call's counter line, not the
        // filling of their pipelines.  With the partial sums stored instead (WfaAlignParams::wave_parts), 100k pairs: 7 iterations 0.089
        // ms, 5 (all that is resident) 0.0715; 50k: 0.064 -> 0.0445; 20k: 0.039 -> 0.022; 250k and more: the same.)
        if (tp.tier == 5 && grid > 0) {
          const uint32_t iters = c->tuning.short_iterations > 0 ? (uint32_t)c->tuning.short_iterations : 1u;
          if (c->tuning.max_blocks_per_cu <= 0) grid = std::min(grid, (int)std::max<uint32_t>(units / iters, 4u * (uint32_t)c->num_cus));
          grid = (int)cdiv(units, cdiv(units, (uint32_t)std::max(grid, 1)));
        }
        // (a speculative re-run works on a list whose length only the device knows yet -- a percent of the chain's pairs,
        // usually: every workgroup beyond the work costs a claim atomic per shard)
        if (cur_len_dev) grid = std::min(grid, (int)std::max<uint32_t>(4u * (uint32_t)c->num_cus, n_cur / 64u));
        // arena refill size: as large as lets every workgroup hold a few chunks -- each refill is a
        // returning atomic on ONE word (~88 per microsecond on this chip), which at 4 KiB refills
        // was the whole kernel time
        // (a refill stalls the whole workgroup for the atomic's round trip, ~2.5 us: at 64 KiB a time -- the cap through round 4 -- an
        // alignment of BASELINE configs[4], 45 MB of origin bytes in rows of up to 13 KB, refilled 700 times, every fifth score, with
        // the CU's only workgroup waiting.  Now: a sixteenth of the arena over the launch's workgroups -- what the launch can
        // strand in unfinished chunks at its end --, at most 4 MiB.)
        if (cigar_now) {
          const unsigned long long cap = c->tuning.arena_chunk_cap > 0 ? (unsigned long long)c->tuning.arena_chunk_cap : 262144ull;
          ap.chunk_units = (uint32_t)std::min<unsigned long long>(cap, std::max<unsigned long long>(256, ap.arena_units / ((cap > 4096 ? 16ull : 4ull) * (unsigned)grid)));
        }
        ap.work_shards = 8u;
        if (zero_counter(c, L.ct_cells, 2)) return -1;   // (the launch's cell count and, next to it, the length of its failure list)
        uint32_t* nxt = spare[flip]; flip ^= 1;
        // (the chain's ONLY launch -- nothing is enqueued behind it that needs its failure list on the device: the list is made by the
        // chain's tail kernel together with the bounds and the count of unfinished pairs)
        fused_tail = n_links == 0 && !(L.budgeted && speculate) && cur_len_dev == nullptr;
        // Tier 5 keeps its cell count as one partial sum per wavefront; whoever runs behind the launch adds them up.  Where the launch
        // is such a chain's only one, score-only, over the whole batch, NOTHING runs behind it: the kernel appends its failures to the
        // next list itself and counts the pairs it does not finish, the partial sums come to the host with the counters (a tail
        // kernel of 13 us + the gap in front of it, of a 135 us step of 100k configs[1] pairs).
        // And where no kernel of this call has touched a counter yet (no pack kernel in front: the launch is the call's first kernel), the
        // wavefronts store their partial sums straight into pinned host memory and the counter block is not copied at all: the chain is
        // one kernel and a stream synchronisation.  (A 57 KB copy behind the kernel started 16 us after it had ended.)
        // With CIGARs the backtrace follows the launch; the kernel between them (failure list, bounds, unfinished count: 16 us for 100k
        // pairs, + the one-thread kernel that moved the arena's bump pointer) goes the same way: the bounds of the scratch come from
        // the chain's budget (by_bound, below).
        ap.wave_parts = nullptr; ap.fail_list = nullptr; ap.fail_count = nullptr; ap.arena_top_known = 0; ap.arena_top0_value = 0;
        parts_n = 0; solo = false; parts_on_host = false; skip_read = false;
        if (tp.tier == 5 && grid > 0 && (size_t)grid <= CT_PARTS_PER_CU * (size_t)c->num_cus) {
          ap.wave_parts = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->counters.p) + CT_PARTS_OFF);
          parts_n = (uint32_t)grid;
          solo = fused_tail && unfiltered && round == 0 && n_cur == n;
          parts_on_host = solo && !c->tuning.no_host_parts;
          skip_read = parts_on_host && !cigar_now && ct_clean_before && (prepacked || fused_pack);
          if (solo) { ap.fail_list = nxt; ap.fail_count = ct + L.ct_list; }
          if (parts_on_host) ap.wave_parts = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->h_counters) + CT_PARTS_OFF);
        }
        // (a pass's first launch finds the arena's bump pointer at zero: zero_counter(CT_ARENA) above)
        if (tp.tier == 5 && cigar_now && round == 0) {
          ap.arena_top_known = 1; ap.arena_top0_value = 0;
          slots_fit = (unsigned long long)n_cur * wfa_short_bt_slot_units(ap.max_score, tp.wpe) <= ap.arena_units;
        }
        // (L.e0 / L.e1: start and end of the wavefront kernel, stamped by its own dispatch packet)
        if (tp.tier == 5) {
          wfa_launch_short(ap, tp.wpe, cigar_now, grid, st, L.e0, L.e1);
          if (solo && !cigar_now) call_end = L.e1;      // (the chain's last kernel)
          // (with CIGARs the launch owns n_cur slots above the bump pointer: move it past them for whatever allocates next)
          if (cigar_now && !ap.arena_top_known) LAUNCH_K(k_bump, dim3(1), dim3(64), 0, st, ap.arena_top, (unsigned long long)n_cur * wfa_short_bt_slot_units(ap.max_score, tp.wpe), ap.arena_units);
        }
        else wfa_launch_align(ap, tp.tier, cigar_now, raw, grid, st, tp.wpe, L.e0, L.e1);
        {
          const hipError_t le = hipGetLastError();
          if (le != hipSuccess) {
            fprintf(stderr, "[!] ERROR: wavefront launch failed: %s (tier %d, grid %d, LDS %zu bytes, %u pairs, score limit %d, window %d, band %d, cigar %d, raw %d)\n",
                    hipGetErrorString(le), tp.tier, grid, tp.lds, n_cur, max_score, tp.width, ap.band_width, (int)cigar_now, (int)raw);
            return -1;
          }
        }
        tail_out = nxt; tail_count = ct + L.ct_list; tail_parts_out = ct + L.ct_cells;
        if (!fused_tail)
          LAUNCH_K(k_compact, dim3(cdiv(n_cur, compact_block(n_cur))), dim3(compact_block(n_cur)), 0, st, (const uint32_t*)cur, n_cur, cur_len_dev,
                             static_cast<const uint32_t*>(c->status.p), MASK(WFA_ST_BAND) | MASK(WFA_ST_SCORE), nxt, ct + L.ct_list, ap.work_counter,
                             (const unsigned long long*)ap.wave_parts, parts_n, ct + L.ct_cells);
        s_hi = std::max<long long>(s_hi, L.banded ? std::min(max_error, 30000) : max_score);
        ++n_links; ++round;
        // what the failures of this round run with next
        cur = nxt; cur_len_dev = ct + L.ct_list;      // (n_cur stays as the upper bound until the chain's synchronisation)
        if (L.banded) {
          // banded misses: exact tiers from the start (under the tuned budgets first, where there are any)
        } else if (budget_round) {
          budget_round = false; max_score = max_error;      // auto-budget misses: the caller's budget, exact tiers
        } else {
          // widen: 4x the score budget (and with it the diagonal window); beyond what 16-bit offsets
          // allow the last resort is the unbounded 32-bit tier
          if (tp.tier == 3 && max_score == INT_MAX) max_score = -1;       // (nothing wider exists: failures are an error, below)
          else if (max_score >= 30000 || max_len > 32766u) max_score = INT_MAX;
          else max_score = (int)std::min<long long>(30000, std::max<long long>(16, 4ll * max_score));   // (a budget of 0 must grow too)
        }
        // the re-run of budget misses follows without a round trip; every other escalation waits for the counts.  (speculate: off
        // when the last batch of the stream had no miss at all -- short reads under budgets with room: an empty launch, its
        // compaction and the gaps around them are 17 us of a 190 us BASELINE configs[1] step; should a pair miss after all, the
        // next chain re-runs it after this one's synchronisation)
        if (!(L.budgeted && n_links == 1 && speculate)) break;
      }
      // (whatever the budget, no optimal alignment costs more than mismatching the shorter sequence and one gap for the rest)
      s_hi = std::min<long long>(s_hi, (long long)pen.x * max_len + oe + (long long)pen.e * max_len);
      // ---- backtrace + CIGAR for everything of the chain's list that finished -----------------------------------------
      if (zero_counter(c, CT_SUM_OPS, 2)) return -1;
      if (zero_counter(c, CT_OPS)) return -1;
      if (zero_counter(c, CT_MAX_SCORE)) return -1;
      bool traced = false;      // (did the backtrace launch anything: are ev_t0 / ev_t1 this chain's)
      if (zero_counter(c, CT_UNFIN)) return -1;
      // (ev_end -- the end of the chain on the device -- is the end stamp of the chain's LAST kernel: this one in score-only calls,
      // the compaction of the NOMEM pairs behind the backtrace otherwise)
      const hipEvent_t ev_tail = cigar_now ? (hipEvent_t) nullptr : c->ev_end;
      if (!(solo && !cigar_now)) call_end = c->ev_end;
      if (solo) {
        // (nothing: the launch did its own bookkeeping)
      } else if (fused_tail)
        LAUNCH_K_END(k_chain_tail, dim3(std::min<uint32_t>(std::max(cdiv(n_chain, 2048), cdiv(n, 8192)), 256u)), dim3(256), 0, st, ev_tail, (const uint32_t*)chain_list, n_chain,
                               static_cast<const uint32_t*>(c->status.p), MASK(WFA_ST_BAND) | MASK(WFA_ST_SCORE), tail_out, tail_count, ap.work_counter,
                               (const int32_t*)d_scores, static_cast<const uint32_t*>(c->cells.p), std::min(pen.x, pen.e), item_chars, ct, n,
                               (const unsigned long long*)ap.wave_parts, parts_n, tail_parts_out);
      else
        LAUNCH_K_END(k_trace_bounds, dim3(std::min<uint32_t>(std::max(cdiv(n_chain, 2048), cdiv(n, 8192)), 256u)), dim3(256), 0, st, ev_tail, (const uint32_t*)chain_list, n_chain,
                               static_cast<const uint32_t*>(c->status.p), (const int32_t*)d_scores,
                               static_cast<const uint32_t*>(c->cells.p), std::min(pen.x, pen.e), item_chars, ct, n);
      if (cigar_now) {
        WfaTraceParams tp{};
        tp.raw = raw ? 1 : 0;
        // both sequences of a pair, as words, per lane; staged in LDS when 64 lanes fit 48 KiB
        const unsigned per_seq = raw ? (max_len + 3) / 4 + 1 : (max_len + 15) / 16 + 1;
        const unsigned stride = (2 * per_seq) | 1u;
        tp.seq_lds_stride = ((size_t)64 * stride * 4 <= (48u << 10) && c->tuning.trace_mode != 1) ? (int)stride : 0;
        // alignments per wavefront of the staged replay: as many as still leave a CU eight wavefronts (two per SIMD).  LDS decides:
        // 1 kbp reads at 64 per wavefront are 4 wavefronts per CU, one per SIMD, every dependent LDS read of the replay exposed;
        // measured on BASELINE configs[2] (trace ms per 1M pairs): 64: 4.14, 56: 3.92, 48: 3.84, 40: 3.84, 32: 3.79, 24: 3.82
        tp.emit_pairs = 64;
        if (tp.seq_lds_stride > 0) {
          while (tp.emit_pairs > 16 && c->lds_per_block_max / ((size_t)tp.emit_pairs * stride * 4) < 8) tp.emit_pairs -= 8;
          const int forced = c->tuning.emit_pairs;
          if (forced >= 8 && forced <= 64) tp.emit_pairs = forced & ~7;
        }
        // sequences too long to stage 64 pairs per wavefront: one wavefront per alignment (sequences of ONE pair in LDS)
        tp.seq_words_cap = (int)per_seq + 1;
        // (it needs both sequences of a pair plus ~1 KiB in LDS; sequences at the very edge of what the align tiers stage do
        // not leave that: the lane-per-alignment walk + windowed emit take any length)
        tp.wave_kernel = (tp.seq_lds_stride == 0 && c->tuning.trace_mode != 1 &&
                          (size_t)2 * tp.seq_words_cap * 4 + 64 * 32 + 64 <= c->lds_per_block_max) ? 1 : 0;      // (64 * 32: the tile, trace_kernel.hip TILE_LDS_W)
        // Scratch sizes.  Lane-per-alignment kernels with a bounded score: from the bound (no pair of the chain finished
        // above s_hi), no round trip.  The wave kernels size their LDS from the largest score of the pass and the
        // unbounded tier has no bound: those read the sums k_trace_bounds has just formed.
        const unsigned long long ops_bound = (unsigned long long)n_chain * (((unsigned long long)std::max<long long>(s_hi, 0) + 3ull) & ~3ull);
        const unsigned long long text_bound = (unsigned long long)n_chain *
            ((unsigned long long)item_chars * (2ull * (unsigned long long)(std::max<long long>(s_hi, 0) / std::min(pen.x, pen.e)) + 1ull) + 1ull);
        // (the bound is sized by the chain's largest budget -- the caller's max_error once a re-run link follows -- for EVERY pair:
        // fine for the batches of a pipelined call, whose round trips it saves (62 500 x 1 kbp pairs: 225 MB), too generous for a
        // big resident batch (1M pairs at max_error 300: 3.6 GB of fresh device memory per buffer to save one counter read)
        const bool by_bound = !tp.wave_kernel && s_hi >= 0 && s_hi <= 30000 && text_bound <= ((unsigned long long)768 << 20);
        unsigned long long ops_need, text_sum;
        if (by_bound) { ops_need = ops_bound + 256; text_sum = text_bound; }
        else {
          // (a chain without a tail kernel: nobody has formed the sums yet.  Cells and the unfinished count stay with the launch's
          // partial sums)
          if (solo) LAUNCH_K(k_trace_bounds, dim3(std::min<uint32_t>(std::max(cdiv(n_chain, 2048), 1u), 256u)), dim3(256), 0, st, (const uint32_t*)chain_list, n_chain,
                             static_cast<const uint32_t*>(c->status.p), (const int32_t*)d_scores, static_cast<const uint32_t*>(nullptr), std::min(pen.x, pen.e), item_chars, ct, 0u);
          if (read_counters(c)) return -1;
          ops_need = c->h_counters[CT_SUM_OPS] + 256; text_sum = c->h_counters[CT_SUM_TEXT];
        }
        const unsigned long long text_need = text_used + text_sum + 256;
        if (c->ops.ensure(ops_need, st)) return -1;
        if (c->text[c->out_set].ensure(std::max<size_t>(text_need, c->text_cfg), st, text_used)) return -1;
        if (!c->text[c->out_set ^ 1].p && c->text[c->out_set ^ 1].ensure(c->text[c->out_set].cap, st)) return -1;      // (the next call's set: see above)
        {
          // op list (one byte per score point of the largest score of the pass) and CIGAR text of one alignment in LDS,
          // within 40 KiB per wavefront so that at least four of them fit a CU
          const size_t seq_b = (size_t)2 * tp.seq_words_cap * 4 + 2048;      // (+ the tile: 64 rows of 32 bytes)
          size_t room = seq_b < (40u << 10) ? (40u << 10) - seq_b : 0;
          const size_t smax = by_bound ? (size_t)s_hi : (size_t)c->h_counters[CT_MAX_SCORE];
          tp.ops_lds_bytes = (int)(((smax + 3) & ~(size_t)3) <= room ? ((smax + 15) & ~(size_t)15) : 0);
          room -= (size_t)tp.ops_lds_bytes;
          tp.text_lds_bytes = (int)(std::min<size_t>(room, 2 * smax + 64) & ~(size_t)15);
          // Several alignments per wavefront (64/G lanes and one LDS share each) for SHORT op lists only (2 kbp reads, G = 8:
          // 262k pairs 13.2 -> 6.6 ms).  With hundreds of operations per alignment the replays of the groups diverge and
          // the wavefront per alignment wins again (16k x 10 kbp: 2.7 ms against 4.0 ms with G = 4).
          const size_t share = seq_b - 2048 + (size_t)tp.ops_lds_bytes + (size_t)tp.text_lds_bytes;
          tp.group = 1;
          if (c->tuning.trace_mode == 0 && tp.ops_lds_bytes > 0 && smax <= 512)
            for (int g = 8; g >= 2; g >>= 1)
              if ((share + (size_t)(64 / g) * 32 + 16) * g <= (40u << 10) && n_chain >= (uint32_t)g * 1024u) { tp.group = g; break; }
        }
        // LONG alignments, one wavefront each: the wavefront kernel only walks; the replay is the staged lane-per-alignment kernel's
        // with as many pairs per wavefront (8, 16, ...) as fit LDS next to three more wavefronts (a replay is one serial chain:
        // run by a whole wavefront it is 64 lanes executing the same chain).  Op lists in slots of the largest score of the pass.
        tp.walk_only = 0;
        const unsigned long long slot = ((unsigned long long)c->h_counters[CT_MAX_SCORE] + 3ull) & ~3ull;
        const size_t per_pair = (size_t)stride * 4 + (size_t)slot + 8;      // LDS of one alignment of the replay: sequences + op list
        // (big passes only: a replay step takes a lane of the lane kernel -- its neighbours in other branches -- 1.3 us against the 0.4 us
        // of a wavefront with a SIMD to itself; with thousands of alignments the wavefronts share SIMDs five at a time and the lanes
        // win: 16 384 x 10 kbp at 3 %, trace 2.22 -> 2.04 ms.  tuning.trace_mode 4: whatever the size of the pass -- tests)
        if (tp.wave_kernel && tp.group == 1 && c->tuning.trace_mode != 2 && (n_chain >= 8192u || c->tuning.trace_mode == 4) &&
            8 * per_pair <= c->lds_per_block_max) {
          if ((unsigned long long)n_chain * slot <= (1ull << 30)) {
            if (c->ops.ensure((unsigned long long)n_chain * slot + 256, st)) return -1;
            tp.walk_only = 1; tp.ops_slot = (uint32_t)slot;
            c->stats.pairs_trace_split += n_chain;
            tp.seq_lds_stride = (int)stride;
            tp.emit_pairs = 8;
            while (tp.emit_pairs < 64 && (size_t)(tp.emit_pairs + 8) * per_pair * 4 <= c->lds_per_block_max) tp.emit_pairs += 8;
            const int forced = c->tuning.emit_pairs;
            if (forced >= 8 && forced <= 64 && (size_t)(forced & ~7) * per_pair <= c->lds_per_block_max) tp.emit_pairs = forced & ~7;
            // (the walk keeps its op list in LDS when it fits: ops_lds_bytes from above, without the sequences)
            const size_t smax = (size_t)c->h_counters[CT_MAX_SCORE];
            tp.ops_lds_bytes = (int)(smax + 16 <= (39u << 10) ? ((smax + 15) & ~(size_t)15) : 0);
            tp.text_lds_bytes = 0;
          }
        }
        // SHORT alignments (no pair of the chain finished above a score of 124): walk and replay in one kernel, op lists in LDS
        tp.lane_fused = 0;
        if (!tp.wave_kernel && tp.seq_lds_stride > 0 && s_hi >= 0 && s_hi <= 124 && c->tuning.trace_mode != 3) {
          tp.lane_fused = 1;
          tp.ops_lds_bytes = (int)((s_hi + 3) & ~3ll);
          if (!(c->tuning.emit_pairs >= 8 && c->tuning.emit_pairs <= 64)) {
            tp.emit_pairs = 64;
            while (tp.emit_pairs > 16 && c->lds_per_block_max / ((size_t)tp.emit_pairs * (stride * 4 + (size_t)tp.ops_lds_bytes) + 64 * 72) < 8) tp.emit_pairs -= 8;
          }
          // (the kernel's workgroups are four wavefronts with a share of LDS each: trace_kernel.hip, LANE_WAVES)
          while (tp.emit_pairs > 8 && 4 * ((size_t)tp.emit_pairs * (stride * 4 + (size_t)tp.ops_lds_bytes) + 64 * 72 + 16) > c->lds_per_block_max) tp.emit_pairs -= 8;
        }
        tp.packed = ap.packed; tp.meta = ap.meta; tp.work = chain_list; tp.n_work = n_chain;
        tp.x = pen.x; tp.oe = oe; tp.e = pen.e;
        tp.score = d_scores; tp.status = static_cast<const uint32_t*>(c->status.p);
        tp.score_fix = (want_band && !raw) ? d_scores : nullptr;
        tp.arena = ap.arena; tp.arena_bytes = (unsigned long long)c->arena.cap; tp.bt_final_row = ap.bt_final_row;
        tp.ops = static_cast<uint8_t*>(c->ops.p); tp.ops_cap = c->ops.cap; tp.ops_top = ct + CT_OPS;
        tp.text = static_cast<char*>(c->text[c->out_set].p); tp.text_cap = c->text[c->out_set].cap; tp.text_top = ct + CT_TEXT;
        tp.min_op_cost = std::min(pen.x, pen.e);
        tp.item_chars = item_chars;
        // lane-per-alignment emit with whole sequences staged: one replay into a scratch + compaction (big passes only: the
        // scratch holds the upper bounds, ~3x the text)
        // (long alignments replayed by the lane kernel -- walk_only -- always: a second replay of hundreds of operations costs more than
        // the compaction launch)
        if (((!tp.wave_kernel && n_chain >= 8192u) || tp.walk_only) && !tp.lane_fused && tp.seq_lds_stride > 0) {
          if (c->text_scratch.ensure(text_sum + 4096, st)) return -1;
          if (zero_counter(c, CT_SCRATCH)) return -1;
          tp.text_scratch = static_cast<char*>(c->text_scratch.p); tp.text_scratch_cap = c->text_scratch.cap; tp.scratch_top = ct + CT_SCRATCH;
        }
        tp.cigar_off = static_cast<unsigned long long*>(c->cig_off[c->out_set].p);
        tp.cigar_len = static_cast<uint32_t*>(c->cig_len[c->out_set].p);
        // (ev_t0 / ev_t1: start of the first and end of the last backtrace kernel)
        tp.walk_grid_cap = 2 * c->num_cus;      // (two workgroups = eight wavefronts per CU: trace_kernel.hip, wfa_walk_kernel)
        // (every link of the chain walked its own alignments: no wfa_walk_kernel; a mixed chain runs it, and it skips the walked pairs)
        tp.skip_walk = 1;
        for (int l = 0; l < n_links; ++l) if (!link[l].walked) tp.skip_walk = 0;
        if (tp.skip_walk) c->stats.pairs_walked_in_kernel += n_chain;
        traced = wfa_launch_trace(tp, st, c->ev_t0, c->ev_t1);
        HIP_TRY(hipGetLastError());
        // ---- pairs that ran out of arena go into the next pass (CIGAR calls only: score-only calls have no arena) ----
        // (none can where a tier-5 launch was the chain's only one and its slots fit the arena: the backtrace kernel is the chain's last
        // then, one launch and the gap in front of it fewer -- 9 of the 165 us of a step of 100k configs[1] pairs with CIGARs)
        if (solo && slots_fit && traced) call_end = c->ev_t1;
        else
        LAUNCH_K_END(k_compact, dim3(cdiv(n_chain, compact_block(n_chain))), dim3(compact_block(n_chain)), 0, st, c->ev_end, (const uint32_t*)chain_list, n_chain, (const unsigned long long*)nullptr,
                               static_cast<const uint32_t*>(c->status.p), MASK(WFA_ST_NOMEM), nxt_pending, ct + CT_NOMEM, (unsigned int*)nullptr,
                               (const unsigned long long*)nullptr, 0u, (unsigned long long*)nullptr);
      }
      // ---- the chain's one synchronisation --------------------------------------------------------------------------
      // (with it comes the number of pairs of the whole batch that are not finished yet: when this turns out to have been the
      // call's last chain, that IS the end-of-call check -- no launch and no round trip of its own)
      // (k_trace_bounds above counted them)
      if (skip_read) {
        HIP_TRY(hipStreamSynchronize(st));
        memset(c->h_counters, 0, CT_N * sizeof(unsigned long long));      // (what the device's counters held before the launch)
      } else if (read_counters(c, (solo && !parts_on_host) ? parts_n : 0u)) return -1;
      if (solo) {
        // the launch's partial sums: cells (of the launch, and of the call: kept on the host, the device's counter never sees them), the
        // pairs of the whole batch that are not finished, the length of the failure list, the pairs flagged for the byte-compare class
        const unsigned long long* parts = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(c->h_counters) + CT_PARTS_OFF);
        unsigned long long sum[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < parts_n; ++i) for (int q = 0; q < 4; ++q) sum[q] += parts[4 * i + q];
        c->h_counters[link[0].ct_cells] += sum[0];
        cells_host += sum[0];
        c->h_counters[CT_UNFIN] = sum[1];
        if (skip_read) {
          c->h_counters[link[0].ct_list] = sum[2]; c->h_counters[CT_NRAW] = sum[3];
          // (the device's counters are as the call found them unless a pair failed or was flagged)
          if (sum[2] == 0 && sum[3] == 0) ct_clean = true;
        }
      }
      unfinished_at_sync = (long long)c->h_counters[CT_UNFIN];
      uint32_t n_in = n_chain;
      for (int l = 0; l < n_links; ++l) {
        const Link& L = link[l];
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, L.e0, L.e1));
        align_ms += ms;
        c->stats.align_launches++;
        if (ms > c->stats.main_launch_ms) {
          c->stats.main_launch_ms = ms; c->stats.main_launch_tier = L.tier; c->stats.main_launch_pairs = n_in;
          c->stats.main_launch_cells = c->h_counters[L.ct_cells];
          c->stats.main_launch_seq_bytes = (unsigned long long)((double)b->packed_bytes * n_in / n);   // (share of the batch)
        }
        const uint32_t n_out = (uint32_t)c->h_counters[L.ct_list];
        // (an unfiltered list: the pairs of the byte-compare class were skipped, not finished)
        const uint32_t skipped = (unfiltered && L.first_round) ? (uint32_t)std::min<unsigned long long>(c->h_counters[CT_NRAW], n_in - n_out) : 0u;
        c->stats.pairs_tier[L.tier == 6 ? 1 : L.tier] += n_in - n_out - skipped;      // (tier 6, two waves, banded only: counted with the four-wave tier)
        if (L.banded) c->stats.pairs_banded += n_in - n_out;
        if (L.first_round) c->stats.pairs_retried += n_out;
        if (L.budgeted) c->stats.pairs_budget_missed += n_out;
        n_in = n_out;
      }
      if (cigar_now) {
        float ms = 0.f;
        if (traced) HIP_TRY(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
        trace_ms += ms;
        text_used = c->h_counters[CT_TEXT];
      }
      n_cur = n_in;      // failures of the chain's last round: the next chain's list (`cur`), now with its exact length
      if (n_cur > 0 && max_score < 0) {
        fprintf(stderr, "[!] ERROR: %u alignments did not finish in the unbounded tier\n", n_cur);
        return -1;
      }
    }
    const uint32_t n_nomem = cigar_now ? (uint32_t)c->h_counters[CT_NOMEM] : 0u;
    if (n_nomem) {
      // grow the arena for the next pass if memory allows; if a pass made no
      // progress at all, also halve the number of alignments in flight
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      const size_t limit = std::min<size_t>((size_t)(0.6 * (double)(free_b + c->arena.cap)), ((size_t)1 << 36) - 4096);
      const size_t grown = std::min<size_t>(limit, std::max<size_t>(2 * c->arena.cap, c->arena.cap + ((size_t)256 << 20)));
      const bool can_grow = !c->arena_cfg && grown > c->arena.cap;
      if (can_grow) {
        if (c->arena.ensure(grown, st)) return -1;
        ap.arena = static_cast<uint8_t*>(c->arena.p);
        ap.arena_units = c->arena.cap / 16 - 8;      // (128 bytes of slack: the walk reads row-table entries a line at a time)
      }
      if (n_nomem == n_pass && !can_grow) {
        if (grid_cap == 1) {
          fprintf(stderr, "[!] ERROR: backtrace arena (%zu bytes) too small for one alignment\n", c->arena.cap);
          return -1;
        }
        const uint32_t cur_cap = std::min<uint32_t>(grid_cap, std::min<uint32_t>(n_pass, (uint32_t)c->num_cus * 32u));
        grid_cap = std::max<uint32_t>(1u, cur_cap / 2);
      }
      LAUNCH_K(k_set_pending, dim3(cdiv(n_nomem, 256)), dim3(256), 0, st, (const uint32_t*)nxt_pending, n_nomem,
                         static_cast<uint32_t*>(c->status.p));
    }
    c->stats.arena_units = std::max<unsigned long long>(c->stats.arena_units, c->h_counters[CT_ARENA]);
    arena_units_call += std::min<unsigned long long>(c->h_counters[CT_ARENA], ap.arena_units);      // (over all passes)
    // refine the per-pair estimate from this pass, then queue what was not launched behind the re-runs
    if (cigar_now && n_pass - n_nomem >= 65536u)   // (few pairs per workgroup: refill slack would dominate)
      est_pair_bytes = std::max(256.0, 1.15 * 16.0 * (double)c->h_counters[CT_ARENA] / (double)(n_pass - n_nomem));
    if (n_pending > n_pass) {
      if (pending) HIP_TRY(hipMemcpyAsync(nxt_pending + n_nomem, pending + n_pass, (size_t)4 * (n_pending - n_pass), hipMemcpyDeviceToDevice, st));
      else LAUNCH_K(k_iota, dim3(cdiv(n_pending - n_pass, 256)), dim3(256), 0, st, nxt_pending + n_nomem, n_pass, n_pending - n_pass);
    }
    // (the list of the next pass is explicit, but still unfiltered where this one was: see only_pending above)
    pending = nxt_pending; n_pending = n_nomem + (n_pending - n_pass);
  }
  return 0;
  };

  // two classes of pairs: ACGT-only (2-bit packed kernels) and the rest (byte-compare kernels)
  for (int cls = 0; cls < 2; ++cls) {
  const bool raw = cls == 1;
  // (the first pass of the packed class has synchronised with the device: the number of flagged pairs is known)
  if (raw && c->h_counters[CT_NRAW] == 0) break;
  max_len = batch_max_len;
  if (raw) ap.packed = reinterpret_cast<const uint32_t*>(b->d_sequences);
  ap.seq_words_cap = raw ? (int)((max_len + 3) / 4 + 1) : (int)((max_len + 15) / 16 + 1);
  const uint32_t class_mask = raw ? MASK(WFA_ST_ALPHABET) : MASK(WFA_ST_PENDING);
  // Length buckets (the reference assumes the pairs of a batch have similar lengths).
  // LDS staging, ring width and with them the tier and the residency are sized by the longest pair that is
  // launched together, so a batch that spans more than 4x in length runs in buckets of 4x each
  // (<= 1024, <= 4096, ...): a few long reads no longer shrink the residency of all the short ones.
  unsigned bucket_lo = 0;
  for (unsigned bucket_hi = batch_max_len > 4096u ? 1024u : batch_max_len; ; bucket_hi = bucket_hi * 4u) {
    if (bucket_hi >= batch_max_len / 2u || bucket_hi > (1u << 30)) bucket_hi = batch_max_len;   // last bucket takes the rest
    uint32_t* const bucket_list = static_cast<uint32_t*>(c->list_c.p);
    uint32_t* pending = bucket_list;
    uint32_t n_pending = n;
    // One bucket, packed class, and no sample to draw (budgets inherited from an earlier batch of the stream, or none to
    // be tuned): the bucket is the identity list, UNFILTERED -- no compaction, no round trip before the first wavefront
    // launch, which skips the pairs flagged for the byte-compare class (their number arrives with the pass's
    // synchronisation).
    bool unfiltered = false;
    if (!raw && bucket_lo == 0 && bucket_hi >= batch_max_len) {
      const int w_me = window_width(max_error, pen.o, pen.e, bucket_hi);
      // (tuned budgets pay where they narrow a wide window -- or, score-only, where they bring it down to what the
      // several-alignments-per-wavefront tier holds)
      const bool short_ok = wfa_short_supported(pen.x, oe, pen.e) && !c->tuning.min_tier && !(compute_cigar && c->tuning.no_short_cigar);
      const bool would_tune = n >= 8192 && !c->tuning.no_auto_budget && !(want_band && c->tuning.force_band) && (w_me > 128 || (w_me > 15 && short_ok));
      bool inherited = false;
      if (would_tune && c->same_stream)
        for (int i = 0; i < c->n_saved_q; ++i) {
          const auto& sq = c->saved_q[i];
          if (sq.bucket_hi == wfagpu_amd_ctx::length_class(bucket_hi) && sq.x == pen.x && sq.o == pen.o && sq.e == pen.e && sq.max_error == max_error) inherited = true;
        }
      unfiltered = !would_tune || inherited;
    }
    if (unfiltered) {
      pending = nullptr;
    } else {
      if (zero_counter(c, CT_LIST)) return -1;
      ct_clean = false;
      if (ensure_status()) return -1;
      LAUNCH_K(k_compact_len, dim3(cdiv(n, 1024)), dim3(1024), 0, st, n, static_cast<const uint32_t*>(c->status.p), class_mask,
                         ap.meta, bucket_lo, bucket_hi, pending, ct + CT_LIST);
      if (read_counters(c)) return -1;
      n_pending = (uint32_t)c->h_counters[CT_LIST];
    }
    if (raw) c->stats.pairs_raw += n_pending;
    if (n_pending) {
      max_len = bucket_hi;
      ap.seq_words_cap = raw ? (int)((max_len + 3) / 4 + 1) : (int)((max_len + 15) / 16 + 1);
      // ---- auto-tuned score budgets ---------------------------------------
      // max_error is a ceiling the caller guesses (the CLI default is 10 % of the length times the largest
      // penalty); the scores of a batch usually sit far below it.  A strided sample of the bucket is aligned
      // with the caller's budget, the 98th percentile of score/length sets a per-pair budget for everyone
      // else, and whoever exceeds it is re-run with the caller's budget.  Results are exact either way; a
      // tight budget halves both the LDS ring and the number of wavefront cells (the wavefront becomes a
      // diamond).
      const int32_t* budgets = nullptr;
      int budget_cap = max_error;
      // Quantile of the sampled score/length ratios, margin on it (percent) and additive slack of the per-pair budgets.
      // A pair that misses its budget is re-run with the caller's (~3x the cells on BASELINE configs[2]), a budget that is
      // too generous costs every pair: 1M x 1 kbp @ 5 %, (quantile, margin, slack) -> budget, misses, align ms:
      // (0.98, 102, 8) 185, 66, 26.19; (0.99, 100, 2) 177, 3.8 k, 25.84; (0.98, 100, 2) 175, 8.6 k, 25.97;
      // (0.95, 100, 2) 171, 27.8 k, 26.35 -- flat around the 99th percentile (a narrower diamond only pays where it
      // saves a whole 64-lane chunk).
      constexpr double budget_q = 0.99;
      constexpr int budget_margin = 100, budget_slack = 2;
      // (with a band requested the sample still runs, exactly: the budgets serve the banded kernels too -- see the band policy below)
      const int w_me2 = window_width(max_error, pen.o, pen.e, max_len);
      const bool short_ok2 = wfa_short_supported(pen.x, oe, pen.e) && !c->tuning.min_tier && !(compute_cigar && c->tuning.no_short_cigar);
      // (the budgets decide whether the band is worth using and serve the exact kernels; with the band forced neither applies)
      const bool try_budget = !raw && n_pending >= 8192 && !c->tuning.no_auto_budget && !(want_band && c->tuning.force_band) &&
                              (w_me2 > 128 || (w_me2 > 15 && short_ok2));
      int saved_idx = -1;
      if (try_budget && c->same_stream) {
        for (int i = 0; i < c->n_saved_q; ++i) {
          const auto& sq = c->saved_q[i];
          if (sq.bucket_hi == wfagpu_amd_ctx::length_class(bucket_hi) && sq.x == pen.x && sq.o == pen.o && sq.e == pen.e && sq.max_error == max_error) saved_idx = i;
        }
      }
      if (try_budget && saved_idx >= 0) {
        // a later batch of the same stream of reads: try the budgets the sample of an earlier batch gave (misses are
        // re-run with the caller's budget as always; too many of them and the next batch samples again)
        const int q = c->saved_q[saved_idx].q;
        const int slack = budget_slack;
        if (c->budget.ensure((size_t)4 * n, st)) return -1;
        budget_rule = {true, q, slack, budget_margin};      // (filled by the first launch that reads it: run_list)
        budgets = static_cast<const int32_t*>(c->budget.p);
        budget_cap = (int)std::min<long long>(max_error, ((long long)q * max_len * budget_margin / 100) / 1024 + slack);
        c->stats.auto_budget = budget_cap;
      } else if (try_budget) {
        const uint32_t n_s = std::min<uint32_t>(4096u, std::max<uint32_t>(512u, n_pending / 16u)), stride_s = n_pending / n_s;
        if (c->sample.ensure((size_t)4 * n_s, st)) return -1;
        if (c->ratio.ensure((size_t)4 * n_s, st)) return -1;
        if (c->budget.ensure((size_t)4 * n, st)) return -1;
        if (c->list_e.ensure((size_t)4 * n, st)) return -1;
        LAUNCH_K(k_sample, dim3(cdiv(n_s, 256)), dim3(256), 0, st, (const uint32_t*)pending, n_s, stride_s, static_cast<uint32_t*>(c->sample.p));
        // (leftovers of the sampled run ping-pong between list_d and list_e; list_c keeps the bucket)
        // The sample only has to yield scores: it runs without backtrace (no arena, no trace launches) and its pairs stay
        // in the bucket -- 0.4 % of the batch aligned twice is cheaper than a separate backtrace pass and a list compaction.
        // (a small sample only: one that is more than 1/64 of the batch keeps its alignments)
        const bool sample_again = compute_cigar && (unsigned long long)n_s * 64ull <= n_pending;
        // (what a sample that is aligned again adds to the work accounting is reported on its own: stats.sample_*)
        wfagpu_amd_stats_t stats_before{};
        unsigned long long cells_before = 0;
        if (sample_again) {
          cigar_now = false;
          if (read_counters(c)) return -1;
          cells_before = c->h_counters[CT_CELLS];
          stats_before = c->stats;
        }
        const int rc_s = run_list(static_cast<uint32_t*>(c->sample.p), n_s, raw, nullptr, max_error, static_cast<uint32_t*>(c->list_d.p),
                                  static_cast<uint32_t*>(c->list_e.p), /*allow_band=*/false);
        cigar_now = compute_cigar;
        if (rc_s) return -1;
        LAUNCH_K(k_ratio, dim3(cdiv(n_s, 256)), dim3(256), 0, st, static_cast<const uint32_t*>(c->sample.p), n_s,
                           static_cast<const uint32_t*>(c->status.p), (const int32_t*)d_scores, ap.meta, static_cast<int32_t*>(c->ratio.p));
        std::vector<int32_t> hr(n_s);
        HIP_TRY(hipMemcpyAsync(hr.data(), c->ratio.p, (size_t)4 * n_s, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (sample_again) {
          if (read_counters(c)) return -1;
          const int launches = c->stats.align_launches - stats_before.align_launches;
          const unsigned passes = c->stats.sub_batches - stats_before.sub_batches;
          c->stats = stats_before;
          c->stats.sample_launches += launches;
          c->stats.sample_passes += passes;
          c->stats.sample_cells += c->h_counters[CT_CELLS] - cells_before;
          sample_cells_call += c->h_counters[CT_CELLS] - cells_before;
        }
        std::sort(hr.begin(), hr.end());
        const size_t valid = std::lower_bound(hr.begin(), hr.end(), INT_MAX) - hr.begin();
        if (valid >= n_s / 2) {
          const int q = hr[std::min(valid - 1, (size_t)(budget_q * valid))];      // score per 1024 bases
          const int slack = budget_slack;
          LAUNCH_K(k_budget, dim3(cdiv(n, 256)), dim3(256), 0, st, ap.meta, n, q, slack, budget_margin, static_cast<int32_t*>(c->budget.p));
          budgets = static_cast<const int32_t*>(c->budget.p);
          budget_cap = (int)std::min<long long>(max_error, ((long long)q * max_len * budget_margin / 100) / 1024 + slack);
          c->stats.auto_budget = budget_cap;
          // remember it for later batches of the same stream
          int slot = -1;
          for (int i = 0; i < c->n_saved_q; ++i) if (c->saved_q[i].bucket_hi == wfagpu_amd_ctx::length_class(bucket_hi)) slot = i;
          if (slot < 0 && c->n_saved_q < 8) slot = c->n_saved_q++;
          if (slot >= 0) c->saved_q[slot] = {wfagpu_amd_ctx::length_class(bucket_hi), q, pen.x, pen.o, pen.e, max_error, 0xFFFFFFFFu};      // (misses under it: not known yet)
        }
        if (sample_again) {
          // the sampled pairs go through the bucket's run like everybody else
          LAUNCH_K(k_set_pending, dim3(cdiv(n_s, 256)), dim3(256), 0, st, static_cast<const uint32_t*>(c->sample.p), n_s, static_cast<uint32_t*>(c->status.p));
        } else {
          // the sampled pairs are done: drop them from the bucket's list
          uint32_t* rest = static_cast<uint32_t*>(c->list_e.p);
          if (zero_counter(c, CT_LIST)) return -1;
          LAUNCH_K(k_compact, dim3(cdiv(n_pending, compact_block(n_pending))), dim3(compact_block(n_pending)), 0, st, (const uint32_t*)pending, n_pending, (const unsigned long long*)nullptr,
                             static_cast<const uint32_t*>(c->status.p), class_mask, rest, ct + CT_LIST);
          if (read_counters(c)) return -1;
          n_pending = (uint32_t)c->h_counters[CT_LIST];
          HIP_TRY(hipMemcpyAsync(pending, rest, (size_t)4 * n_pending, hipMemcpyDeviceToDevice, st));
        }
      }
      // Band policy: the band is a permission to approximate, not an obligation.  Where the tuned budgets leave the exact
      // wavefronts narrower than 1.75 bands the exact search costs no more than the band does and every result is optimal: the
      // band is not used for the bucket (tuning.force_band keeps it).  Otherwise, and in batches too small to tune, the
      // banded kernels run.  (The threshold was 2.5 bands through round 4; since round 5 the banded kernels run a cell as fast
      // as the exact ones and need no budget re-runs: BASELINE configs[3] -- windows of 1.94 bands -- 16.0 ms exact against
      // 14.0 ms banded per step.  The exact diamond has ~0.45 W cells per score, the band ~0.92 beta: even at W = 1.75 beta.)
      const bool use_band = want_band && (!budgets || c->tuning.force_band ||
                                          4 * window_width(budget_cap, pen.o, pen.e, max_len) > 7 * band_width);
      const unsigned missed_before = c->stats.pairs_budget_missed;
      const bool speculate = !(saved_idx >= 0 && c->saved_q[saved_idx].last_missed == 0);
      if (n_pending && run_list(pending, n_pending, raw, budgets, budget_cap, static_cast<uint32_t*>(c->list_d.p), bucket_list, use_band, speculate)) return -1;
      budget_rule.unset = false;
      if (saved_idx >= 0) c->saved_q[saved_idx].last_missed = c->stats.pairs_budget_missed - missed_before;
      if (saved_idx >= 0 && (c->stats.pairs_budget_missed - missed_before) * 20u > n_pending) {
        // more than 5 % of the batch missed the inherited budgets: the stream has drifted, sample again next time
        c->saved_q[saved_idx] = c->saved_q[--c->n_saved_q];
      }
    }
    if (bucket_hi >= batch_max_len) break;
    bucket_lo = bucket_hi + 1u;
  }
  }  // class loop
  if (pen_scale > 1) LAUNCH_K(k_scale_scores, dim3(cdiv(n, 256)), dim3(256), 0, st, d_scores, n, pen_scale);
  // every pair must have been finished by one of the lists above; anything else is a driver bug and must not
  // be returned as a result
  if (unfinished_at_sync < 0 || pen_scale > 1) {
    // (something was queued behind the last chain's synchronisation -- list bookkeeping of a pass that turned out to be the
    // last, the score scaling: count again and drain the stream, the call is blocking)
    if (zero_counter(c, CT_UNFIN)) return -1;
    LAUNCH_K(k_count_unfinished, dim3(std::min<uint32_t>(cdiv(n, 1024), 1024u)), dim3(256), 0, st, n, static_cast<const uint32_t*>(c->status.p), ct + CT_UNFIN);
    HIP_TRY(hipEventRecord(c->ev_end, st));
    call_end = c->ev_end;
    ct_clean = false;
    if (read_counters(c)) return -1;
    unfinished_at_sync = (long long)c->h_counters[CT_UNFIN];
  }
  if (unfinished_at_sync != 0) {
    fprintf(stderr, "[!] ERROR: %lld of %u alignments were left unfinished\n", unfinished_at_sync, n);
    return -1;
  }
  c->stats.cells = c->h_counters[CT_CELLS] + cells_host - sample_cells_call;
  float ms = 0.f;
  if (!prepacked && !fused_pack) HIP_TRY(hipEventElapsedTime(&ms, c->ev_start, c->ev_pack));
  c->stats.pack_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev_start, call_end)); c->stats.total_ms = ms;
  // several arena-bound passes under a growable cap: the next call may use twice the arena
  if (compute_cigar && c->arena_limit && c->arena_limit_max > c->arena_limit && c->stats.sub_batches > 1 &&
      c->arena.cap >= c->arena_limit - ((size_t)1 << 20))
    // (straight to what this call used over all its passes, if that is more than twice the cap: every re-allocation
    // of a multi-GiB arena is paid for again when the fresh memory is first written)
    c->arena_limit = std::min(c->arena_limit_max, std::max<size_t>(2 * c->arena_limit, (size_t)((double)arena_units_call * 16.0 * 1.1)));
  c->stats.auto_budget *= pen_scale;
  c->stats.align_ms = align_ms;
  c->stats.trace_ms = trace_ms;
  c->stats.text_bytes = text_used;
  if (compute_cigar) {
    if (d_text) *d_text = static_cast<const char*>(c->text[c->out_set].p);
    if (d_off) *d_off = static_cast<const unsigned long long*>(c->cig_off[c->out_set].p);
    if (d_len) *d_len = static_cast<const unsigned int*>(c->cig_len[c->out_set].p);
    c->out_set ^= 1;      // (the next call writes the other set)
  }
  if (rc == 0 && (ct_clean || hipMemsetAsync(c->counters.p, 0, CT_BYTES + 8 * 64, st) == hipSuccess)) c->counters_zeroed = true;
  return rc;
}
