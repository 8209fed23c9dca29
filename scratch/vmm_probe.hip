// Does a device buffer that GROWS IN PLACE work on this box, and what does it cost?  hipMemAddressReserve + hipMemCreate / hipMemMap /
// hipMemSetAccess in granules against hipMalloc: time of each call, fill bandwidth over the mapped range, growth after a large free
// (the driver wipes released memory in the background; a large hipMalloc behind a large hipFree stalls: profiles/r04/cold_long.txt).
//   hipcc -O2 --offload-arch=gfx950 scratch/vmm_probe.hip -o scratch/vmm_probe.bin && scratch/vmm_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(uint4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
static double fill_ms(void* p, size_t bytes) {
  hipDeviceSynchronize();
  const double t0 = now_ms();
  hipLaunchKernelGGL(fill, dim3(256 * 16), dim3(256), 0, 0, (uint4*)p, bytes / 16);
  hipDeviceSynchronize();
  return now_ms() - t0;
}

int main() {
  OK(hipSetDevice(0));
  OK(hipFree(0));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  OK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity (recommended) %zu bytes\n", gran);
  const size_t GiB = (size_t)1 << 30;
  const size_t reserve = 96 * GiB;
  void* base = nullptr;
  double t0 = now_ms();
  OK(hipMemAddressReserve(&base, reserve, 0, nullptr, 0));
  printf("reserve 96 GiB of addresses: %.3f ms -> %p\n", now_ms() - t0, base);
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> handles;
  size_t mapped = 0;
  auto grow = [&](size_t bytes, const char* what) -> int {
    const double a = now_ms();
    hipMemGenericAllocationHandle_t h;
    OK(hipMemCreate(&h, bytes, &prop, 0));
    const double b = now_ms();
    OK(hipMemMap((char*)base + mapped, bytes, 0, h, 0));
    const double c = now_ms();
    OK(hipMemSetAccess((char*)base + mapped, bytes, &acc, 1));
    const double d = now_ms();
    handles.push_back(h); mapped += bytes;
    printf("%s: +%zu GiB: create %.3f ms, map %.3f, access %.3f -> %zu GiB mapped\n", what, bytes / GiB, b - a, c - b, d - c, mapped / GiB);
    return 0;
  };
  // (a single map / set-access of 4 GiB and more fails with "invalid argument" on this runtime: granules of at most 2 GiB)
  if (grow(1 * GiB, "granule")) return 1;
  if (grow(1 * GiB, "granule")) return 1;
  for (int i = 0; i < 11; ++i) if (grow(2 * GiB, "granule")) return 1;
  printf("fill 24 GiB mapped range: first %.2f ms", fill_ms(base, mapped));
  printf(", second %.2f ms\n", fill_ms(base, mapped));
  void* plain = nullptr;
  t0 = now_ms();
  OK(hipMalloc(&plain, 24 * GiB));
  printf("hipMalloc 24 GiB: %.3f ms; ", now_ms() - t0);
  printf("fill first %.2f ms", fill_ms(plain, 24 * GiB));
  printf(", second %.2f ms\n", fill_ms(plain, 24 * GiB));
  // a large free, then growth right behind it: in place against a fresh hipMalloc
  t0 = now_ms();
  OK(hipFree(plain));
  printf("hipFree 24 GiB (touched): %.3f ms\n", now_ms() - t0);
  for (int i = 0; i < 4; ++i) if (grow(2 * GiB, "right after the free")) return 1;
  t0 = now_ms();
  OK(hipMalloc(&plain, 30 * GiB));
  printf("hipMalloc 30 GiB right after: %.3f ms\n", now_ms() - t0);
  printf("fill the 8 GiB just mapped: %.2f ms\n", fill_ms((char*)base + mapped - 8 * GiB, 8 * GiB));
  OK(hipFree(plain));
  t0 = now_ms();
  OK(hipMemUnmap(base, mapped));
  for (auto h : handles) OK(hipMemRelease(h));
  OK(hipMemAddressFree(base, reserve));
  printf("unmap + release + free addresses: %.3f ms\n", now_ms() - t0);
  return 0;
}
