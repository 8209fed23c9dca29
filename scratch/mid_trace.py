"""Mid-length reads (2 kbp @ 3 %): the several-alignments-per-wavefront backtrace (wfa_trace_group_kernel).  usage: mid_trace.py [pairs]"""
import sys, time
sys.path.insert(0, "wfa-gpu_amd/bindings"); sys.path.insert(0, "tests")
import numpy as np, wfagpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
buf, meta = wfagpu.generate_pairs(n, 2000, 0.03, seed=5)
al = wfagpu.DeviceAligner(0)
al.hint_same_stream(True)
batch = al.upload(buf, meta)
ref = None
for i in range(6):
    s, c = al.align(batch, (2, 3, 1), max_error=400, compute_cigar=True)
    st = al.stats()
    if ref is None: ref = (s.copy(), c)
    else: assert np.array_equal(s, ref[0]) and c == ref[1]
    print(f"call {i}: total {st.total_ms:.3f} ms align {st.align_ms:.3f} trace {st.trace_ms:.3f} tiers {list(st.pairs_tier)}", flush=True)
import zlib
print("checksum", zlib.crc32(("".join(ref[1])).encode()) & 0xffffffff, int(ref[0].sum()))
al.close()
