"""The adaptive band this build ships (band_cut, csrc/align_kernel.hip) against the REFERENCE's window rule restated on the CPU
(oracle/band_oracle.c: lib/kernels/sequence_distance_kernel_aband.cu:91-130), on the same pairs, same beta / lambda / step
limit: share of pairs that finish inside the band, share that finish inside the band WITH the optimal score, mean excess of
the others.  Pairs a band does not finish are not counted as hits here (this build re-runs them exactly on the GPU, the
reference hands them to WFA2's adaptive heuristic on the CPU: either way they are not the band's merit).
Writes gpurun_out/band_rules.md (-> profiles/r04/banded.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib

PEN = (2, 3, 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NT = min(32, len(os.sched_getaffinity(0)))
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max": NT = max(1, min(NT, int(int(q) / int(p))))
except Exception:
    pass
hifi = wfagpu.read_seq_file(os.path.join(ROOT, "tests", "golden", "test_hifi.seq"))
sets = [
    ("iid 10 kbp, 3 % single-base edits (generate_dataset model)", lambda: wfagpu.generate_pairs(N, 10000, 0.03, seed=5, nthreads=16), 3000, (352, 512, 1024)),
    ("long-read shaped 10 kbp: 6 % events, 60 % indels (geometric mean 2.5, 2 % long 30-150 bp), 30 % clustered",
     lambda: wfagpu.generate_pairs_model(N, 10000, seed=6, error=0.06, indel_frac=0.6, indel_mean=2.5, long_frac=0.02, long_min=30,
                                         long_max=150, cluster=0.3, nthreads=16), 6000, (352, 512, 1024)),
    ("long-read shaped 10 kbp, heavier: 10 % events, 70 % indels (mean 3, 5 % long 50-400 bp), 50 % clustered",
     lambda: wfagpu.generate_pairs_model(N, 10000, seed=7, error=0.10, indel_frac=0.7, indel_mean=3.0, long_frac=0.05, long_min=50,
                                         long_max=400, cluster=0.5, nthreads=16), 14000, (352, 512, 1024)),
    ("the reference's tests/data/test_hifi.seq, all 50 pairs (real HiFi-shaped reads, 6-19 kbp)", lambda: wfagpu.layout_pairs(hifi), 3000, (32, 64, 128, 352)),
]
LAMS = (10, 25, 100)
al = wfagpu.DeviceAligner(0, force_band=1)
out = ["# Adaptive band: this build's rule (band_cut) against the reference's rule restated on the CPU", "",
       f"Penalties (2,3,1); {N} pairs per synthetic set; truth = the exact GPU run (bit-exact vs WFA2 in tests/).  `inside` = pairs that finish inside",
       "the band within the step limit; `optimal` = pairs that finish inside the band with the optimal score (share of ALL pairs);",
       "`excess` = mean (banded - optimal) / optimal over the pairs that finish inside the band with a worse score.",
       "reference rule = oracle/band_oracle.c (sequence_distance_kernel_aband.cu:91-130, deterministic restatement); this build = the GPU run,",
       "`tuning.force_band = 1`, score + CIGAR.", ""]
summary = []
for name, gen, me, betas in sets:
    buf, meta = gen()
    n = len(meta)
    batch = al.upload(buf, meta)
    d_s, _ = al.align(batch, PEN, max_error=me, compute_cigar=True, fetch=False)
    exact = d_s.cpu().numpy().copy()
    out += [f"## {name}", "", f"{n} pairs, step limit (-e) {me}; optimal score: mean {exact.mean():.0f}, max {exact.max()}", "",
            "| beta | lambda | ref inside | ref optimal | ref excess | this build inside | this build optimal | this build excess |", "|---|---|---|---|---|---|---|---|"]
    for beta in betas:
        for lam in LAMS:
            t0 = time.time()
            sr = oracle_lib.band_ref_batch(buf, meta, PEN, beta, lam, me, nthreads=NT)
            t_ref = time.time() - t0
            fin = sr >= 0
            assert (sr[fin] >= exact[fin]).all()
            r_in = fin.mean(); r_opt = (fin & (sr == exact)).mean()
            bad = fin & (sr != exact)
            r_exc = ((sr[bad] - exact[bad]) / exact[bad]).mean() if bad.any() else 0.0
            d_s, _ = al.align(batch, PEN, max_error=me, compute_cigar=True, band=lam, band_width=beta, fetch=False)
            st = al.stats()
            s = d_s.cpu().numpy()
            assert (s >= exact).all()
            inside = int(st.pairs_banded)
            m_in = inside / n; m_opt = (int((s == exact).sum()) - (n - inside)) / n
            badm = s != exact
            m_exc = ((s[badm] - exact[badm]) / exact[badm]).mean() if badm.any() else 0.0
            out.append(f"| {beta} | {lam} | {r_in * 100:.2f} % | {r_opt * 100:.2f} % | {r_exc * 100:.2f} % | {m_in * 100:.2f} % | {m_opt * 100:.2f} % | {m_exc * 100:.2f} % |")
            summary.append((name[:20], beta, lam, r_in, r_opt, m_in, m_opt))
            print(out[-1], f"  (oracle {t_ref:.1f} s)", flush=True)
    out.append("")
open(os.path.join(ROOT, "gpurun_out", "band_rules.md"), "w").write("\n".join(out) + "\n")
