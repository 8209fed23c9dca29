"""Why is launch_alignments 4 ms slower inside bench.py than in scratch/hostpath.py on the same box?  Variants of the process state."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
import numpy as np
import wfagpu
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
n = 1_000_000
lib = wfagpu.load()
buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000, nthreads=16)
if mode in ("torch", "torch_resident"):
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(1, device="cuda")
if mode == "torch_resident":
    al = wfagpu.DeviceAligner(0)
    batch = al.upload(buf, meta)
    for _ in range(3):
        al.align(batch, (2, 3, 1), max_error=300, compute_cigar=True, fetch=False)
    al.close(); del batch; torch.cuda.empty_cache()
if mode == "resident_own_stream":
    import torch
    al = wfagpu.DeviceAligner(0, use_torch_stream=False)
    batch = al.upload(buf, meta)
    for _ in range(3):
        al.align(batch, (2, 3, 1), max_error=300, compute_cigar=True, fetch=False)
    al.close(); del batch; torch.cuda.empty_cache()
res = C.POINTER(wfagpu.AlignmentResult)()
assert lib.initialize_wfa_results(C.byref(res), n, 256)
opt = wfagpu.Options(max_error=300, threads_per_block=64, num_workers=0, band=-1, batch_size=n, num_alignments=n,
                     penalties=wfagpu.Penalties(2, 3, 1), compute_cigar=True)
ms = []
for r in range(8):
    t0 = time.perf_counter()
    lib.launch_alignments(buf.ctypes.data, buf.nbytes, meta.ctypes.data, res, opt, False)
    ms.append((time.perf_counter() - t0) * 1e3)
st = wfagpu.last_launch_stats()
print(mode, " ".join(f"{m:.1f}" for m in ms), "| d2h %.1f device %.1f upload %.1f scatter %.1f" % (st["d2h_ms"], st["device_ms"], st["upload_ms"], st["scatter_ms"]), flush=True)
