"""The one-wave exact kernels compiled for 8 / 7 waves per SIMD on reads whose rings are small enough for 32 per CU (400 bp @ 5 %: windows beyond
tier 5's 32 lanes): which instantiation should the host pick when LDS allows 29..32 rings?   scratch/wpe_probe.py [pairs] [length] [error]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 400
err = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
buf, meta = wfagpu.generate_pairs(n, length, err, seed=1000, nthreads=16)
for pen, me in (((2, 3, 1), int(length * 0.3)), ((5, 3, 2), int(length * 0.6)), ((3, 1, 4), int(length * 0.7))):
    for wpe in (0, 7, 8):
        al = wfagpu.DeviceAligner(0, waves_per_simd=wpe); batch = al.upload(buf, meta)
        al.align(batch, pen, max_error=me, compute_cigar=True, fetch=False); al.hint_same_stream(True)
        al.align(batch, pen, max_error=me, compute_cigar=True, fetch=False)
        best = None
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            al.align(batch, pen, max_error=me, compute_cigar=True, fetch=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            st = al.stats()
            if best is None or dt < best[0]: best = (dt, st.main_launch_ms, st.main_launch_tier, st.blocks_per_cu_tier0, st.waves_per_simd_tier0, st.auto_budget)
        print(f"{length} bp @ {err}: penalties {pen} waves_per_simd={wpe}: step {best[0]*1e3:.3f} ms main {best[1]:.3f} ms tier {best[2]} rings/CU {best[3]} compiled for {best[4]} budget {best[5]}", flush=True)
        al.close()
