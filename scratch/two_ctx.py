"""Does the backtrace of one half of a batch hide under the wavefront kernel of the other?  Two contexts (own streams),
each with part of the cfg3 batch, driven from two host threads; compared with one context on the whole batch.
  python scratch/two_ctx.py [parts] [max_blocks_per_cu]"""
import sys, os, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
import numpy as np, torch, wfagpu
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bpc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
slices = int(sys.argv[3]) if len(sys.argv) > 3 else 4      # sub-batches per context
n = 1_000_000
buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000, nthreads=16)
def run(parts, bpc, slices):
    als, batches = [], []
    per = n // (parts * slices)
    for p in range(parts):
        al = wfagpu.DeviceAligner(0, use_torch_stream=False, max_blocks_per_cu=bpc)
        als.append(al)
        bl = []
        for s in range(slices):
            i0 = (p * slices + s) * per
            m = meta[i0:i0 + per].copy()
            lo = int(min(m["pattern_offset"].min(), m["text_offset"].min())); hi = int(max((m["pattern_offset"] + m["pattern_len"]).max(), (m["text_offset"] + m["text_len"]).max())) + 8
            m["pattern_offset"] -= lo; m["text_offset"] -= lo
            bl.append(al.upload(buf[lo:hi], m))
        batches.append(bl)
    def work(p, reps):
        for _ in range(reps):
            for b in batches[p]:
                als[p].align(b, (2, 3, 1), max_error=300, compute_cigar=True, fetch=False)
    for reps, tag in ((2, "warm"), (5, "timed")):
        th = [threading.Thread(target=work, args=(p, reps)) for p in range(parts)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"parts {parts} slices {slices} bpc {bpc}: {dt*1e3:.2f} ms per 1M pairs", flush=True)
    for al in als: al.close()
run(1, 0, 1)
run(parts, bpc, slices)
