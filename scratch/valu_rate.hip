// Microbenchmark: issue rate of VALU instruction classes per SIMD on gfx950 (what "peak" means for a kernel that is bound by
// vector instruction issue).  Every mode runs 16 x 4 independent chains of ONE instruction (inline asm: the compiler cannot
// fuse or drop them) in every lane of 256 x 4 x W one-wave workgroups (W waves per SIMD), and reports cycles per wave
// instruction per SIMD at the clock the device reports for the run (wall_clock64-free: hipEvent time x nominal 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int* out, int iters, int seed) {
  int a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 5, d = c + 7;
  float fa = (float)a, fb = (float)b, fc = (float)c, fd = (float)d;
  int s = seed;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) { a = a + b; b = b ^ c; c = max(c, d); d = d - a; }            // 4 int VALU, compiler's choice
      if (MODE == 1) { a = a + b; b = b ^ c; c = max(c, d); d = d - a; s = s * 5 + 1; s ^= j; }  // + 2 SALU
      if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
      if (MODE == 3) asm volatile("v_pk_add_u16 %0, %0, %1\n\tv_pk_add_u16 %1, %1, %2\n\tv_pk_add_u16 %2, %2, %3\n\tv_pk_add_u16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 4) asm volatile("v_pk_max_i16 %0, %0, %1\n\tv_pk_max_i16 %1, %1, %2\n\tv_pk_max_i16 %2, %2, %3\n\tv_pk_max_i16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 5) asm volatile("v_max3_i32 %0, %0, %1, %2\n\tv_max3_i32 %1, %1, %2, %3\n\tv_max3_i32 %2, %2, %3, %0\n\tv_max3_i32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 6) asm volatile("v_alignbit_b32 %0, %0, %1, %2\n\tv_alignbit_b32 %1, %1, %2, %3\n\tv_alignbit_b32 %2, %2, %3, %0\n\tv_alignbit_b32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 7) asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n\tv_lshl_add_u32 %1, %1, 1, %2\n\tv_lshl_add_u32 %2, %2, 1, %3\n\tv_lshl_add_u32 %3, %3, 1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 8) asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 9) asm volatile("v_max_i32 %0, %0, %1\n\tv_max_i32 %1, %1, %2\n\tv_max_i32 %2, %2, %3\n\tv_max_i32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 10) asm volatile("v_bfi_b32 %0, %0, %1, %2\n\tv_bfi_b32 %1, %1, %2, %3\n\tv_bfi_b32 %2, %2, %3, %0\n\tv_bfi_b32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 11) asm volatile("v_ffbl_b32 %0, %1\n\tv_ffbl_b32 %1, %2\n\tv_ffbl_b32 %2, %3\n\tv_ffbl_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 12) asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1" : "+v"(*(double*)&fa), "+v"(*(double*)&fc));
      if (MODE == 13) asm volatile("v_min3_i32 %0, %0, %1, %2\n\tv_med3_i32 %1, %1, %2, %3\n\tv_xor_b32 %2, %2, %3\n\tv_sub_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if (MODE == 14) asm volatile("v_pk_sub_i16 %0, %0, %1\n\tv_pk_min_i16 %1, %1, %2\n\tv_pk_lshlrev_b16 %2, 1, %3\n\tv_pk_max_u16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + s + (int)(fa + fb + fc + fd);
}
template <int MODE> void run(const char* name, int waves_per_simd, int per_iter = 4) {
  int* out; hipMalloc(&out, 1 << 26);
  const int blocks = 256 * 4 * waves_per_simd;   // 64-thread blocks
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 16, 1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double valu = (double)blocks * iters * 16 * per_iter;
  printf("%-28s waves/SIMD %d: %7.3f ms, %.2f G wave-instr, %.2f cycles per instruction per SIMD @2.4GHz\n", name, waves_per_simd, ms,
         valu / 1e9, 1024 * (ms * 1e-3 * 2.4e9) / valu);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) run<0>("int add/xor/max/sub", w);
  for (int w : {4, 8}) run<1>("same + 2 SALU per 4 VALU", w);
  for (int w : {1, 2, 4, 8}) run<2>("v_fma_f32", w);
  for (int w : {1, 2, 4, 8}) run<12>("v_pk_fma_f32", w, 2);
  for (int w : {1, 2, 4, 8}) run<3>("v_pk_add_u16", w);
  for (int w : {1, 2, 4, 8}) run<4>("v_pk_max_i16", w);
  for (int w : {1, 2, 4, 8}) run<14>("v_pk_sub/min/lshl/max 16", w);
  for (int w : {1, 2, 4, 8}) run<5>("v_max3_i32", w);
  for (int w : {1, 2, 4, 8}) run<6>("v_alignbit_b32", w);
  for (int w : {1, 2, 4, 8}) run<7>("v_lshl_add_u32", w);
  for (int w : {1, 2, 4, 8}) run<8>("v_add_u32", w);
  for (int w : {1, 2, 4, 8}) run<9>("v_max_i32", w);
  for (int w : {1, 2, 4, 8}) run<10>("v_bfi_b32", w);
  for (int w : {1, 2, 4, 8}) run<11>("v_ffbl_b32", w);
  for (int w : {1, 2, 4, 8}) run<13>("v_min3/med3/xor/sub", w);
  return 0;
}
