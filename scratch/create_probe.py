import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
import torch, wfagpu
torch.cuda.set_device(0); x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
hip = wfagpu._hiprt()
for name, fn in (("hipStreamCreateWithFlags", lambda: hip.hipStreamCreateWithFlags(C.byref(C.c_void_p()), 1)),
                 ("hipEventCreate", lambda: hip.hipEventCreate(C.byref(C.c_void_p()))),
                 ("hipHostMalloc 4 KB", lambda: hip.hipHostMalloc(C.byref(C.c_void_p()), 4096, 0)),
                 ("hipHostMalloc 16 MB", lambda: hip.hipHostMalloc(C.byref(C.c_void_p()), 16 << 20, 0)),
                 ("hipMalloc 1 MB", lambda: hip.hipMalloc(C.byref(C.c_void_p()), 1 << 20)),
                 ("hipMalloc 4 GB", lambda: hip.hipMalloc(C.byref(C.c_void_p()), 4 << 30))):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:28s}", " ".join(f"{t:.2f}" for t in ts), "ms")
for i in range(3):
    t0 = time.perf_counter(); al = wfagpu.DeviceAligner(0, use_torch_stream=False); t1 = time.perf_counter()
    print(f"wfagpu_amd_create #{i}: {(t1-t0)*1e3:.2f} ms")
