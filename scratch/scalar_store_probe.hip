#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// Does gfx950 still have scalar stores (s_store_dwordx2), do they reach memory before the next kernel reads, and do dirty lines of
// the scalar cache merge by byte with what other wavefronts -- scalar or vector stores, other CUs -- write into the same 64-byte line?
// Layout: consecutive 8-byte slots; slot i is written by block (i * 7919) % nb: even slots by a scalar store, odd slots by a vector
// store of lane 0 -- neighbours in a line always come from different blocks and different kinds of store.
__global__ void k(unsigned long long* out, int n_slots, int nb) {
  for (int i = 0; i < n_slots; ++i) {
    if ((int)(((long long)i * 7919) % nb) != (int)blockIdx.x) continue;
    const unsigned long long v = 0xABCD000000000000ull | (unsigned long long)i;
    if (i & 1) { if (threadIdx.x == 0) out[i] = v; }
    else {
      unsigned long long base = (unsigned long long)out; unsigned int off = (unsigned int)i * 8u;
      asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(v), "s"(base), "s"(off) : "memory");
    }
  }
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
int main() {
  const int nb = 2048, n = 1 << 16;
  unsigned long long* d; (void)hipMalloc(&d, (size_t)n * 8);
  unsigned long long* h = (unsigned long long*)malloc((size_t)n * 8);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(d, 0, (size_t)n * 8);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d, n, nb);
    (void)hipMemcpy(h, d, (size_t)n * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) if (h[i] != (0xABCD000000000000ull | (unsigned long long)i)) ++bad;
    printf("interleaved scalar / vector stores from %d blocks: %ld wrong of %d slots\n", nb, bad, n);
  }
  return 0;
}
