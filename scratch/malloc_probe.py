"""How long do hipMalloc / hipFree of arena-sized buffers take on the box?"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize()
for rep in range(2):
    for gb in (1, 4, 8, 16, 32, 44):
        p = C.c_void_p()
        t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), gb << 30); t1 = time.perf_counter()
        hip.hipMemset(p, 0, 1 << 20); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipFree(p); t3 = time.perf_counter()
        print(f"rep {rep}: {gb:3d} GiB: hipMalloc {1e3*(t1-t0):8.1f} ms (rc {rc}), first touch of 1 MiB {1e3*(t2-t1):6.1f} ms, hipFree {1e3*(t3-t2):8.1f} ms", flush=True)
