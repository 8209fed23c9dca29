"""Does the state of the process's device memory (earlier big allocations, freed or kept) change the time of resident steps?
  python scratch/resident_state.py <mode>    mode: fresh | prefree (8 GB hipMalloc + memset + hipFree first) | prekeep (kept)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
import numpy as np, torch, wfagpu
mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
buf, meta = wfagpu.generate_pairs(1_000_000, 1000, 0.05, seed=1000, nthreads=16)
keep = None
if mode in ("prefree", "prekeep"):
    x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda"); x.zero_(); torch.cuda.synchronize()
    if mode == "prefree":
        del x; torch.cuda.empty_cache(); torch.cuda.synchronize()
    else:
        keep = x
al = wfagpu.DeviceAligner(0)
b = al.upload(buf, meta)
for i in range(6):
    if i == 2:
        al.hint_same_stream(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    al.align(b, (2, 3, 1), max_error=300, compute_cigar=True, fetch=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    st = al.stats()
    print(f"{mode} step {i}: {dt:.2f} ms  main {st.main_launch_ms:.2f} align {st.align_ms:.2f} trace {st.trace_ms:.2f} pack {st.pack_ms:.2f}", flush=True)
