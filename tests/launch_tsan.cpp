// ThreadSanitizer harness for the host pipeline of launch_alignments* (csrc/wfa_launch.hip compiled as C++ against the stub HIP
// layer of tests/hip_stub/; built and run by tests/test_sanitizers.py).  Calls of several shapes -- 1 / 3 / 4 lanes, input pools
// that hold every batch or only a ring of them, sequences packed on the host or going up as ASCII, one device and eight virtual
// ones, score-only and CIGAR, with -c -- each checked against the oracle pair by pair.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/wfa_gpu_device.h"
#include "../oracle/wfa_oracle.h"

extern "C" size_t wfagen_pair_stride(int length, double error);
extern "C" size_t wfagen_generate(char* seqbuf, size_t cap, sequence_pair_t* meta, size_t n, int length, double error, uint64_t seed, int nthreads);
#include <hip/hip_runtime.h>      // (the stub layer of tests/hip_stub: hipSetDevice / hipGetDevice / stub_set_device_count)

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)

int main() {
  const size_t n = 6000;
  const int length = 120;
  const size_t cap = wfagen_pair_stride(length, 0.06) * n + 16;
  std::vector<char> seq(cap);
  std::vector<sequence_pair_t> meta(n);
  const size_t used = wfagen_generate(seq.data(), cap, meta.data(), n, length, 0.06, 11, 2);
  CHECK(used != 0);
  // a few pairs with bytes outside ACGT: their batches go up as ASCII even when the host packs
  for (size_t i = 500; i < n; i += 1500) if (meta[i].text_len > 5) seq[meta[i].text_offset + 3] = 'N';
  const affine_penalties_t pen = {2, 3, 1};
  std::vector<int> want(n);
  std::vector<std::vector<char>> want_cg(n);
  oracle_aligner_t* al = oracle_aligner_new(2, 3, 1);
  for (size_t i = 0; i < n; ++i) {
    want_cg[i].resize(1024);
    want[i] = oracle_align(al, seq.data() + meta[i].pattern_offset, (int)meta[i].pattern_len, seq.data() + meta[i].text_offset, (int)meta[i].text_len,
                           want_cg[i].data(), want_cg[i].size(), nullptr);
  }
  oracle_aligner_delete(al);

  struct Shape { int lanes, virt, host_pack; size_t pool, batch; bool cigar, check; };
  const Shape shapes[] = {{1, 0, -1, 0, 6000, true, false}, {3, 0, 1, 0, 400, true, false}, {4, 0, -1, 1 << 16, 250, true, true},
                          {3, 0, 1, 1 << 16, 300, false, false}, {2, 8, -1, 0, 200, true, false}, {3, 8, 1, 1 << 15, 100, true, true},
                          {0, 3, 0, 0, 1000, false, true}};
  for (const Shape& sh : shapes) {
    wfagpu_amd_launch_config_t cfg{};
    cfg.lanes_per_device = sh.lanes; cfg.virtual_devices = sh.virt; cfg.host_pack = sh.host_pack; cfg.host_pack_threads = 2;
    cfg.input_pool_bytes = sh.pool; cfg.bring_up = -1;
    wfagpu_amd_configure_launch(&cfg);
    for (int rep = 0; rep < 2; ++rep) {      // cold (lanes come up under the first batches) and warm
      wfa_alignment_result_t* res = nullptr;
      CHECK(initialize_wfa_results(&res, n, rep ? 64 : 4));      // (4-byte CIGAR buffers: every record grows)
      wfa_alignment_options_t opt{};
      opt.max_error = 60; opt.threads_per_block = 64; opt.band = -1; opt.batch_size = sh.batch; opt.num_alignments = n; opt.penalties = pen;
      opt.compute_cigar = sh.cigar;
      std::vector<sequence_pair_t> m2(meta);
      if (sh.cigar) launch_alignments(seq.data(), used, m2.data(), res, opt, sh.check);
      else launch_alignments_distance(seq.data(), used, m2.data(), res, opt, sh.check);
      for (size_t i = 0; i < n; ++i) {
        CHECK((int)res[i].error == want[i]);
        if (sh.cigar) CHECK(strcmp(res[i].cigar.buffer, want_cg[i].data()) == 0);
      }
      if (sh.check) CHECK(wfagpu_amd_check_failures() == 0);
      wfagpu_amd_launch_stats_t st;
      wfagpu_amd_last_launch_stats(&st);
      CHECK(st.devices == (sh.virt ? sh.virt : 1));
      destroy_wfa_results(res, n);
    }
    wfagpu_amd_release_cache();
  }
  // the background bring-up (first device query) followed at once by a call, and by a release
  wfagpu_amd_configure_launch(nullptr);
  int nd = 0;
  get_num_cuda_devices(&nd);
  CHECK(nd == 1);
  {
    wfa_alignment_result_t* res = nullptr;
    CHECK(initialize_wfa_results(&res, n, 64));
    wfa_alignment_options_t opt{};
    opt.max_error = 60; opt.threads_per_block = 64; opt.band = -1; opt.batch_size = 500; opt.num_alignments = n; opt.penalties = pen; opt.compute_cigar = true;
    std::vector<sequence_pair_t> m2(meta);
    launch_alignments(seq.data(), used, m2.data(), res, opt, false);
    for (size_t i = 0; i < n; ++i) CHECK((int)res[i].error == want[i] && strcmp(res[i].cigar.buffer, want_cg[i].data()) == 0);
    destroy_wfa_results(res, n);
  }
  // the explicit warm-up over several devices leaves the caller's current device where it was (the query functions too)
  stub_set_device_count(4);
  CHECK(hipSetDevice(2) == hipSuccess);
  wfagpu_amd_warmup();
  int cur = -1;
  CHECK(hipGetDevice(&cur) == hipSuccess && cur == 2);
  get_num_cuda_devices(&nd);
  CHECK(nd == 4 && get_cuda_SM_count(0) == 256);
  CHECK(hipGetDevice(&cur) == hipSuccess && cur == 2);
  wfagpu_amd_release_cache();
  stub_set_device_count(1);
  printf("launch_tsan ok\n");
  return 0;
}
