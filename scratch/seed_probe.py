import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
import numpy as np, torch, wfagpu
for seed in (7, 1000, 8):
    buf, meta = wfagpu.generate_pairs(1_000_000, 1000, 0.05, seed=seed, nthreads=16)
    print("seed", seed, "max plen", int(meta["pattern_len"].max()), "max tlen", int(meta["text_len"].max()), "min tlen", int(meta["text_len"].min()))
    al = wfagpu.DeviceAligner(0)
    per = 62500
    for b in range(3):
        m = meta[b * per:(b + 1) * per].copy()
        lo = int(min(m["pattern_offset"].min(), m["text_offset"].min())); hi = int(max((m["pattern_offset"] + m["pattern_len"]).max(), (m["text_offset"] + m["text_len"]).max())) + 8
        m["pattern_offset"] -= lo; m["text_offset"] -= lo
        batch = al.upload(buf[lo:hi], m)
        for rep in range(2):
            torch.cuda.synchronize()
            al.align(batch, (2, 3, 1), max_error=300, compute_cigar=True, fetch=False)
        st = al.stats()
        print("  batch", b, "max len", batch.max_seq_len, "budget", st.auto_budget, "missed", st.pairs_budget_missed, "lds", st.lds_bytes_tier0, "blocks/CU", st.blocks_per_cu_tier0,
              "wpe", st.waves_per_simd_tier0, "cells", st.cells, "align ms %.3f trace %.3f total %.3f main %.3f" % (st.align_ms, st.trace_ms, st.total_ms, st.main_launch_ms))
    al.close()
