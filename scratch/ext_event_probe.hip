// Do the start / stop events of hipExtLaunchKernelGGL time a kernel without the extra barrier packets of hipEventRecord, and can
// the start event of one launch be paired with the stop event of a later one?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out) out[0] = 1;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  int* d; hipMalloc(&d, 4);
  hipEvent_t a0, a1, b0, b1, r0, r1, r2, r3;
  for (hipEvent_t* e : {&a0, &a1, &b0, &b1, &r0, &r1, &r2, &r3}) hipEventCreate(e);
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000LL, d); hipStreamSynchronize(s);
  for (int rep = 0; rep < 3; ++rep) {
    // (wall_clock64 ticks at 100 MHz: 10000 = 100 us, 5000 = 50 us)
    hipEventRecord(r0, s);
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, a0, a1, 0, 10000LL, d);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000LL, d);
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, b0, b1, 0, 5000LL, d);
    hipEventRecord(r1, s);
    hipStreamSynchronize(s);
    float aa, bb, ab, a0b0, rr, r0a0, b1r1;
    hipEventElapsedTime(&aa, a0, a1); hipEventElapsedTime(&bb, b0, b1); hipEventElapsedTime(&ab, a0, b1); hipEventElapsedTime(&a0b0, a0, b0);
    hipEventElapsedTime(&rr, r0, r1); hipEventElapsedTime(&r0a0, r0, a0); hipEventElapsedTime(&b1r1, b1, r1);
    printf("ext: A %.1f us (100)  B %.1f us (50)  A.start->B.stop %.1f us (170+)  A.start->B.start %.1f  | records around: %.1f us, r0->A.start %.1f, B.stop->r1 %.1f\n",
           aa * 1e3, bb * 1e3, ab * 1e3, a0b0 * 1e3, rr * 1e3, r0a0 * 1e3, b1r1 * 1e3);
    // the same with hipEventRecord around every kernel
    hipEventRecord(r0, s);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 10000LL, d);
    hipEventRecord(r1, s);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000LL, d);
    hipEventRecord(r2, s);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 5000LL, d);
    hipEventRecord(r3, s);
    hipStreamSynchronize(s);
    float x, y, z;
    hipEventElapsedTime(&x, r0, r1); hipEventElapsedTime(&y, r2, r3); hipEventElapsedTime(&z, r0, r3);
    printf("rec: A %.1f us  B %.1f us  whole %.1f us\n", x * 1e3, y * 1e3, z * 1e3);
    // only one of the two events given
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, a0, nullptr, 0, 10000LL, d);
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, b1, 0, 5000LL, d);
    hipStreamSynchronize(s);
    hipError_t e = hipEventElapsedTime(&ab, a0, b1);
    printf("start-only + stop-only: %.1f us (150)  rc %d\n", ab * 1e3, (int)e);
  }
  return 0;
}
