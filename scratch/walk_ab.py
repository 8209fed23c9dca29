"""In-kernel walk (round 5) against the walk kernel on BASELINE configs[2]: step, main launch, trace; parity of both on a sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000, nthreads=16)
idx = np.arange(0, n, max(1, n // 4000))
so, co = oracle_lib.ref_batch(buf, meta[idx], (2, 3, 1), cigar=True, memory_mode=0, nthreads=16) if oracle_lib.have_ref() else oracle_lib.oracle_batch(buf, meta[idx], (2, 3, 1), cigar=True, nthreads=16)[:2]
for rnd in range(2):
    for kw in (0, 1):
        al = wfagpu.DeviceAligner(0, kernel_walk=kw)
        batch = al.upload(buf, meta)
        al.align(batch, (2, 3, 1), max_error=300, compute_cigar=True, fetch=False)
        al.hint_same_stream(True)
        ts, ms, tr = [], [], []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = al.align(batch, (2, 3, 1), max_error=300, compute_cigar=True, fetch=False)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            st = al.stats(); ms.append(st.main_launch_ms); tr.append(st.trace_ms)
        st = al.stats()
        sc = out[0].cpu().numpy()
        cg = wfagpu.fetch_cigars(out[1][0], out[1][1], out[1][2], n, st.text_bytes)
        ok = bool(np.array_equal(sc[idx], so)) and all(cg[i] == co[j] for j, i in enumerate(idx))
        print(f"kernel_walk={kw}: step {np.median(ts):.2f} ms main {np.median(ms):.2f} align {st.align_ms:.2f} trace {np.median(tr):.2f} walked {st.pairs_walked_in_kernel} tiers {list(st.pairs_tier)} parity {ok}", flush=True)
        al.close(); del batch; torch.cuda.empty_cache()
