// Microbenchmark: does a wave64 VALU instruction get cheaper on gfx950 when whole 16/32-lane groups of EXEC are off?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int iters, int seed, int active) {
  int a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 5, d = c + 7;
  if ((int)threadIdx.x < active) {     // whole loop under a partial EXEC
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { a = a + b; b = b ^ c; c = max(c, d); d = d - a; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main() {
  int* out; hipMalloc(&out, 1 << 26);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w : {4, 8}) for (int active : {64, 48, 33, 32, 17, 16, 8, 1}) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 16, 1, active);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters, 1, active);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double valu = (double)blocks * iters * 16 * 4;
    printf("waves/SIMD %d, active lanes %2d: %.3f ms => %.2f cycles per VALU wave-instruction per SIMD @2.4GHz\n", w, active, ms,
           1024 * (ms * 1e-3 * 2.4e9) / valu);
  }
  return 0;
}
