"""Throughput of the exact kernels under other penalty sets (1M x 1 kbp @ 5 %, score + CIGAR, resident batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000, nthreads=16)
al = wfagpu.DeviceAligner(0); batch = al.upload(buf, meta)
for pen, me in (((2, 3, 1), 300), ((4, 6, 2), 600), ((1, 2, 1), 200), ((5, 3, 2), 600), ((3, 1, 4), 700)):
    al.align(batch, pen, max_error=me, compute_cigar=True, fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d_s, ptrs = al.align(batch, pen, max_error=me, compute_cigar=True, fetch=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = al.stats()
    k = 500
    so, co, _ = oracle_lib.oracle_batch(buf, meta[:k], pen, cigar=True, nthreads=16)
    cg = wfagpu.fetch_cigars(ptrs[0], ptrs[1], ptrs[2], n, st.text_bytes)[:k]
    ok = np.array_equal(d_s[:k].cpu().numpy(), so) and cg == co
    print(f"penalties {pen}: {dt*1e3:.1f} ms  {n/dt/1e6:.2f} M alignments/s  align {st.align_ms:.1f} trace {st.trace_ms:.1f}  cells {st.cells/1e9:.2f} G  exact {ok}", flush=True)
