"""What would a smaller ring buy?  (a) 30 kbp pairs whose whole ring fits LDS (16-wave tier) against the same pairs forced onto
the hybrid tier (D ring in HBM); (b) 10 kbp pairs on the four-wave tier at 8 rings per CU against the same capped at 6."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu

def run(tag, n, L, err, me, **tun):
    al = wfagpu.DeviceAligner(0, **tun)
    buf, meta = wfagpu.generate_pairs(n, L, err, seed=9, nthreads=16); batch = al.upload(buf, meta)
    al.align(batch, (2, 3, 1), max_error=me, compute_cigar=True, fetch=False)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        al.align(batch, (2, 3, 1), max_error=me, compute_cigar=True, fetch=False)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    st = al.stats()
    print(f"{tag}: n {n} L {L} err {err} e {me} {tun}: {best*1e3:.2f} ms align {st.align_ms:.2f} main {st.main_launch_ms:.2f} tier {st.main_launch_tier} "
          f"tiers {list(st.pairs_tier)} cells {st.main_launch_cells/1e9:.2f}G budget {st.auto_budget}", flush=True)
    del al

for me in (6000, 7000):
    run("30k fits LDS", 1024, 30000, 0.07, me)
    run("30k hybrid  ", 1024, 30000, 0.07, me, min_tier=4)
for me in (700, 800, 1000):
    run("10k 4-wave ", 16384, 10000, 0.02, me, no_auto_budget=1)
    run("10k capped 6", 16384, 10000, 0.02, me, no_auto_budget=1, max_blocks_per_cu=6)
    run("10k capped 5", 16384, 10000, 0.02, me, no_auto_budget=1, max_blocks_per_cu=5)
