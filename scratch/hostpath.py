"""Times launch_alignments* (the reference's host seam: pageable host buffers in, host results out) on a
synthetic batch, the way the CLI's wall-time line does (tools/aligner.c:450-474 of the reference)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
import numpy as np
if os.environ.get("IMPORT_TORCH"):
    import torch
    if os.environ["IMPORT_TORCH"] == "2":
        torch.zeros(1, device="cuda")
import wfagpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
err = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
cigar = (sys.argv[4] != "score") if len(sys.argv) > 4 else True
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
batch = int(sys.argv[6]) if len(sys.argv) > 6 else n
lib = wfagpu.load()
lanes = int(sys.argv[7]) if len(sys.argv) > 7 else 0
nbatch = int(sys.argv[8]) if len(sys.argv) > 8 else 0
bpc = int(sys.argv[9]) if len(sys.argv) > 9 else 0
wfagpu.configure_launch(timing=int(os.environ.get("TIMING", "0")), lanes_per_device=lanes, batches_per_device=nbatch, tuning={"max_blocks_per_cu": bpc},
                        host_pack=int(os.environ.get("HOST_PACK", "0")), host_pack_threads=int(os.environ.get("PACK_THREADS", "0")),
                        ascii_every=int(os.environ.get("ASCII_EVERY", "0")))
buf, meta = wfagpu.generate_pairs(n, length, err, seed=int(os.environ.get("SEED", "7")), nthreads=16)
bl = os.environ.get("BENCHLIKE", "")
if bl and bl[0] == "x":
    hip = wfagpu._hiprt()
    ptr = C.c_void_p()
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    assert hip.hipMalloc(C.byref(ptr), buf.nbytes) == 0
    assert hip.hipMemset(ptr, 0, buf.nbytes) == 0
    hip.hipDeviceSynchronize()
    if bl != "xk":
        assert hip.hipFree(ptr) == 0
elif bl and bl[0] in "tmh":
    import torch
    if bl[0] == "t":
        x = torch.empty(buf.nbytes, dtype=torch.uint8, device="cuda"); x.zero_(); torch.cuda.synchronize(); del x
    elif bl[0] == "m":
        x = torch.from_numpy(buf).to("cuda"); torch.cuda.synchronize(); del x
    else:
        x = torch.empty(buf.nbytes, dtype=torch.uint8, device="cuda")
        hip = wfagpu._hiprt()
        assert hip.hipMemcpy(x.data_ptr(), buf.ctypes.data, buf.nbytes, 1) == 0
        del x
    torch.cuda.empty_cache()
elif bl:
    import torch
    al = wfagpu.DeviceAligner(0)
    if "u" in bl:
        b = al.upload(buf, meta)
        if "a" in bl:
            for _ in range(3):
                al.align(b, (2, 3, 1), max_error=300, compute_cigar=cigar, fetch=False)
        del b
    al.close()
    torch.cuda.empty_cache()
    if "f" in bl:
        buf = buf.copy()
res = C.POINTER(wfagpu.AlignmentResult)()
assert lib.initialize_wfa_results(C.byref(res), n, 256)
opt = wfagpu.Options(max_error=int(length * 0.1 * 3), threads_per_block=64, num_workers=0, band=-1, batch_size=batch,
                     num_alignments=n, penalties=wfagpu.Penalties(2, 3, 1), compute_cigar=cigar)
fn = lib.launch_alignments if cigar else lib.launch_alignments_distance
for r in range(reps):
    t0 = time.perf_counter()
    fn(buf.ctypes.data, buf.nbytes, meta.ctypes.data, res, opt, False)
    dt = time.perf_counter() - t0
    print(f"call {r}: {dt*1e3:.1f} ms  {n/dt/1e6:.2f} M alignments/s  ({'CIGAR' if cigar else 'score'})", flush=True)
    st = wfagpu.last_launch_stats()
    print("   ", " ".join(f"{k}={v:.1f}" if isinstance(v, float) else f"{k}={v}" for k, v in st.items()), flush=True)
print("first:", res[0].error, C.string_at(res[0].cigar.buffer)[:60] if cigar and res[0].cigar.buffer else "")
