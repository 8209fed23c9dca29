// sync_probe.hip -- what the end of a blocking call costs: kernel + hipStreamSynchronize, + a D2H copy of a counter line in between,
// against a kernel that writes the line and a flag into pinned host memory which the host polls.
//   hipcc --offload-arch=gfx950 -O2 scratch/sync_probe.hip -o scratch/sync_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(unsigned long long* ct, int spin) {
  unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < spin) {}
  if (threadIdx.x == 0) atomicAdd(ct, 1ull);
}
__global__ void k_publish(const unsigned long long* ct, volatile unsigned long long* host, unsigned long long seq) {
  if (threadIdx.x < 16) host[threadIdx.x] = ct[threadIdx.x];
  __threadfence_system();
  if (threadIdx.x == 0) { __atomic_store_n((unsigned long long*)&host[16], seq, __ATOMIC_RELEASE); }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long* d; OK(hipMalloc(&d, 256)); OK(hipMemset(d, 0, 256));
  unsigned long long* h; OK(hipHostMalloc(&h, 256, hipHostMallocDefault)); for (int i = 0; i < 32; ++i) h[i] = 0;
  const int N = 2000;
  for (int spin : {0, 10000}) {      // (wall_clock64 ticks at 100 MHz: 10000 = 100 us)
    for (int mode = 0; mode < 4; ++mode) {
      double best = 1e9, sum = 0;
      for (int i = 0; i < N + 50; ++i) {
        const double t0 = now();
        hipLaunchKernelGGL(k_work, dim3(256), dim3(64), 0, st, d, spin);
        if (mode == 0) { OK(hipStreamSynchronize(st)); }
        else if (mode == 1) { OK(hipMemcpyAsync(h, d, 128, hipMemcpyDeviceToHost, st)); OK(hipStreamSynchronize(st)); }
        else if (mode == 2) {
          hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d, h, (unsigned long long)(i + 1 + mode * 100000));
          while (__atomic_load_n(&h[16], __ATOMIC_ACQUIRE) != (unsigned long long)(i + 1 + mode * 100000)) {}
        } else {
          hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d, h, (unsigned long long)(i + 1 + mode * 100000));
          OK(hipStreamSynchronize(st));
        }
        const double t = now() - t0;
        if (i >= 50) { sum += t; if (t < best) best = t; }
      }
      const char* names[4] = {"kernel + sync", "kernel + D2H copy + sync", "kernel + publish kernel, host polls", "kernel + publish kernel + sync"};
      printf("spin %6d  %-38s mean %7.2f us  best %7.2f us\n", spin, names[mode], sum / N, best);
    }
  }
  return 0;
}
