"""Adaptive-band recall/time study (mirrors README.md:125-137 of the reference: beta in {352, 512, 1024} x lambda in
{10, 25, 50, 100, 750}) on two data sets: the i.i.d. single-base errors of WFA2's generate_dataset (easy: no gap moves the
optimal path by more than one diagonal at a time) and long-read shaped pairs with multi-base indels, a few long ones and
clustered errors (tools/generate_dataset.c: wfagen_generate_model).  Ground truth = the exact GPU run of the same library
(itself bit-exact against WFA2 in tests/).  Writes a markdown table to gpurun_out/banded_study.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib

PEN = (2, 3, 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sets = [
    ("iid 10 kbp, 3 % single-base edits (generate_dataset model)", lambda: wfagpu.generate_pairs(N, 10000, 0.03, seed=5, nthreads=16), 3000),
    ("long-read shaped 10 kbp: 6 % events, 60 % indels (geometric mean 2.5, 2 % long 30-150 bp), 30 % clustered",
     lambda: wfagpu.generate_pairs_model(N, 10000, seed=6, error=0.06, indel_frac=0.6, indel_mean=2.5, long_frac=0.02, long_min=30,
                                         long_max=150, cluster=0.3, nthreads=16), 6000),
    ("long-read shaped 10 kbp, heavier: 10 % events, 70 % indels (mean 3, 5 % long 50-400 bp), 50 % clustered",
     lambda: wfagpu.generate_pairs_model(N, 10000, seed=7, error=0.10, indel_frac=0.7, indel_mean=3.0, long_frac=0.05, long_min=50,
                                         long_max=400, cluster=0.5, nthreads=16), 14000),
]
al = wfagpu.DeviceAligner(0)
out = ["# Adaptive band: time and recall against the exact GPU run", "",
       f"{N} pairs per set, penalties (2,3,1), score + CIGAR, one MI355X, batch resident in HBM; time = wall of one call (ms).",
       "recall = share of pairs whose banded score equals the optimum; excess = mean (banded - optimal) / optimal over the others.", ""]
for name, gen, me in sets:
    buf, meta = gen()
    batch = al.upload(buf, meta)
    al.align(batch, PEN, max_error=me, compute_cigar=True, fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d_s, _ = al.align(batch, PEN, max_error=me, compute_cigar=True, fetch=False)
    torch.cuda.synchronize(); t_exact = (time.perf_counter() - t0) * 1e3
    exact = d_s.cpu().numpy().copy()
    # spot check of the truth itself against the checker
    k = 16
    so = oracle_lib.ref_batch(buf, meta[:k], PEN, cigar=False, nthreads=8)[0] if oracle_lib.have_ref() else oracle_lib.oracle_batch(buf, meta[:k], PEN, cigar=False, nthreads=8)[0]
    assert np.array_equal(exact[:k], so)
    out += [f"## {name}", "", f"exact: {t_exact:.1f} ms ({N / t_exact * 1e3:.0f} pairs/s), mean score {exact.mean():.0f}, max {exact.max()}", "",
            "| beta | lambda | ms | pairs/s | vs exact | recall | mean excess | finished inside the band |", "|---|---|---|---|---|---|---|---|"]
    for beta in (352, 512, 1024):
        for lam in (10, 25, 50, 100, 750):
            al.align(batch, PEN, max_error=me, compute_cigar=True, band=lam, band_width=beta, fetch=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d_s, _ = al.align(batch, PEN, max_error=me, compute_cigar=True, band=lam, band_width=beta, fetch=False)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            st = al.stats()
            s = d_s.cpu().numpy()
            assert (s >= exact).all()
            hit = s == exact
            exc = ((s[~hit] - exact[~hit]) / exact[~hit]).mean() if (~hit).any() else 0.0
            out.append(f"| {beta} | {lam} | {dt:.1f} | {N / dt * 1e3:.0f} | {t_exact / dt:.2f}x | {hit.mean() * 100:.2f} % | {exc * 100:.2f} % | {st.pairs_banded} / {N} |")
            print(out[-1], flush=True)
    out.append("")
open(os.path.join(ROOT, "gpurun_out", "banded_study.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
