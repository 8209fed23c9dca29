// Host->device copy rates from pageable, registered and pinned-staged memory (what launch_alignments* can choose from).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 512) << 20;
  char* h = (char*)malloc(bytes); memset(h, 1, bytes);
  char* d; hipMalloc(&d, bytes);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int r = 0; r < 3; ++r) { double t0 = now(); hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("pageable H2D   %.1f ms  %.1f GB/s\n", now() - t0, bytes / (now() - t0) / 1e6); }
  { double t0 = now(); hipHostRegister(h, bytes, hipHostRegisterDefault); printf("hipHostRegister %.1f ms\n", now() - t0); }
  for (int r = 0; r < 3; ++r) { double t0 = now(); hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("registered H2D %.1f ms  %.1f GB/s\n", now() - t0, bytes / (now() - t0) / 1e6); }
  { double t0 = now(); hipHostUnregister(h); printf("hipHostUnregister %.1f ms\n", now() - t0); }
  // staged: nt threads memcpy into pinned chunks, each followed by an async copy
  const size_t chunk = 16u << 20; const int nbuf = 4;
  char* pin[nbuf]; for (auto& p : pin) hipHostMalloc((void**)&p, chunk, hipHostMallocDefault);
  hipEvent_t ev[nbuf]; for (auto& e : ev) hipEventCreate(&e);
  for (int nt : {1, 2, 4, 8}) {
    double t0 = now();
    size_t off = 0; int i = 0;
    while (off < bytes) {
      const size_t n = std::min(chunk, bytes - off);
      if (i >= nbuf) hipEventSynchronize(ev[i % nbuf]);
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { const size_t a = n * t / nt, b = n * (t + 1) / nt; memcpy(pin[i % nbuf] + a, h + off + a, b - a); });
      for (auto& t : th) t.join();
      hipMemcpyAsync(d + off, pin[i % nbuf], n, hipMemcpyHostToDevice, s);
      hipEventRecord(ev[i % nbuf], s);
      off += n; ++i;
    }
    hipStreamSynchronize(s);
    printf("staged %d threads %.1f ms  %.1f GB/s\n", nt, now() - t0, bytes / (now() - t0) / 1e6);
  }
  // D2H into pinned and pageable
  char* hp; hipHostMalloc((void**)&hp, bytes, hipHostMallocDefault);
  for (int r = 0; r < 2; ++r) { double t0 = now(); hipMemcpyAsync(hp, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("D2H pinned %.1f ms %.1f GB/s\n", now() - t0, bytes / (now() - t0) / 1e6); }
  for (int r = 0; r < 2; ++r) { double t0 = now(); hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("D2H pageable %.1f ms %.1f GB/s\n", now() - t0, bytes / (now() - t0) / 1e6); }
  { double t0 = now(); char* p2; hipHostMalloc((void**)&p2, bytes, hipHostMallocDefault); printf("hipHostMalloc %zu MB %.1f ms\n", bytes >> 20, now() - t0); }
  { double t0 = now(); char* d2; hipMalloc(&d2, bytes); printf("hipMalloc %zu MB %.1f ms\n", bytes >> 20, now() - t0); }
  printf("hw threads %u\n", std::thread::hardware_concurrency());
  return 0;
}
