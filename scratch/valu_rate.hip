// Microbenchmark: issue rate of integer VALU / SALU instructions per SIMD on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int* out, int iters, int seed) {
  int a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 5, d = c + 7;
  int s = seed;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) { a = a + b; b = b ^ c; c = max(c, d); d = d - a; }            // 4 independent-ish int VALU
      if (MODE == 1) { a = a + b; b = b ^ c; c = max(c, d); d = d - a; s = s * 5 + 1; s ^= j; }  // + 2 SALU
      if (MODE == 2) { a = __builtin_amdgcn_alignbit(a, b, c); b = a ^ d; c = min(c, b); d = d + 1; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  int* out; hipMalloc(&out, 1 << 26);
  const int blocks = 256 * 4 * waves_per_simd;   // 64-thread blocks
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 16, 1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double valu = (double)blocks * iters * 16 * 4;
  printf("%s waves/SIMD %d: %.3f ms, %.2f G VALU wave-instr, %.3f VALU/cycle/SIMD @2.4GHz (=> %.2f cycles per VALU)\n", name, waves_per_simd, ms,
         valu / 1e9, valu / 1024 / (ms * 1e-3 * 2.4e9), 1024 * (ms * 1e-3 * 2.4e9) / valu);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) run<0>("int add/xor/max/sub", w);
  for (int w : {4, 8}) run<1>("same + 2 SALU per 4 VALU", w);
  for (int w : {4, 8}) run<2>("alignbit/xor/min/add", w);
  return 0;
}
