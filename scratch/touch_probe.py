"""What does "first touch" of fresh device memory cost on this box, and when is it paid?  (hipMalloc returns in < 1 ms whatever
the size; the first kernels that write a fresh backtrace arena run at ~30 GB/s.)
  A  malloc G GiB, memset all at once (first touch), memset again (steady)
  B  malloc, sleep 1.5 s, memset        -- is the cost paid in the background after the allocation?
  C  malloc 8 GiB, memset only the first GiB, then the rest     -- per touched byte, or per allocation?
  D  free, malloc the same size again at once, memset           -- is freed memory cheaper (or dearer) to get back?
  E  two allocations touched one after the other vs. concurrently on two streams"""
import ctypes as C, time, sys
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
GiB = 1 << 30
def malloc(n):
    p = C.c_void_p(); t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); assert rc == 0, rc
    return p, (time.perf_counter() - t0) * 1e3
def memset(p, off, n):
    t0 = time.perf_counter(); hip.hipMemsetAsync(C.c_void_p(p.value + off), 1, n, None); hip.hipDeviceSynchronize()
    return (time.perf_counter() - t0) * 1e3
def free(p):
    t0 = time.perf_counter(); hip.hipFree(p); return (time.perf_counter() - t0) * 1e3
w, _ = malloc(1 << 20); memset(w, 0, 1 << 20)     # runtime + fill kernel warm
for g in (1, 4, 16):
    p, tm = malloc(g * GiB); t1 = memset(p, 0, g * GiB); t2 = memset(p, 0, g * GiB)
    print(f"A {g:2d} GiB: hipMalloc {tm:.2f} ms, first memset {t1:.1f} ms ({t1 / g:.1f} ms/GiB), second {t2:.1f} ms", flush=True)
    tf = free(p)
    p, tm = malloc(g * GiB); t1 = memset(p, 0, g * GiB)
    print(f"D {g:2d} GiB: hipFree {tf:.1f} ms, malloc again {tm:.2f} ms, first memset {t1:.1f} ms ({t1 / g:.1f} ms/GiB)", flush=True)
    free(p)
time.sleep(2.0)
p, tm = malloc(8 * GiB); time.sleep(1.5); t1 = memset(p, 0, 8 * GiB)
print(f"B  8 GiB: malloc, 1.5 s later first memset {t1:.1f} ms ({t1 / 8:.1f} ms/GiB)", flush=True)
free(p); time.sleep(2.0)
p, tm = malloc(8 * GiB); t1 = memset(p, 0, GiB); t2 = memset(p, GiB, 7 * GiB); t3 = memset(p, 0, 8 * GiB)
print(f"C  8 GiB: first GiB {t1:.1f} ms, the other seven {t2:.1f} ms ({t2 / 7:.1f} ms/GiB), all again {t3:.1f} ms", flush=True)
free(p); time.sleep(2.0)
# many small allocations instead of one big one
t0 = time.perf_counter(); ps = [malloc(256 << 20)[0] for _ in range(32)]; tm = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
for q in ps: hip.hipMemsetAsync(q, 1, 256 << 20, None)
hip.hipDeviceSynchronize(); t1 = (time.perf_counter() - t0) * 1e3
print(f"F 32 x 256 MiB: mallocs {tm:.1f} ms, first memsets {t1:.1f} ms ({t1 / 8:.1f} ms/GiB)", flush=True)
for q in ps: free(q)
