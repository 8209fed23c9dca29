"""Reads a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory and prints, for the window of the LAST call of a
hostpath run, the GPU busy time per kernel, the idle gaps between consecutive kernels, and the biggest gaps.
  python scratch/timeline.py <dir> [window_ms]"""
import csv, glob, sys, collections
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
import re
def kname(n):
    m = re.search(r"(wfa_\w+|k_\w+|__amd_rocclr_\w+)(<[^>]*>)?", n)
    return (m.group(0) if m else n)[:80]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
cp = []
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Size", 0) or 0)))
rows.sort()
t_end = rows[-1][1]
t0 = t_end - int(win * 1e6)
sel = [r for r in rows if r[0] >= t0]
print("kernels in the last %.0f ms: %d" % (win, len(sel)))
busy = collections.Counter(); cnt = collections.Counter()
for s, e, k, q in sel: busy[k] += e - s; cnt[k] += 1
for k, v in busy.most_common(25): print("  %9.3f ms  %5d x  %s" % (v / 1e6, cnt[k], k))
# union of busy intervals
iv = sorted((s, e) for s, e, _, _ in sel)
tot = 0; cur_s, cur_e = iv[0]
gaps = []
for s, e in iv[1:]:
    if s > cur_e: gaps.append((s - cur_e, cur_e)); tot += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
tot += cur_e - cur_s
print("span %.2f ms, GPU busy (union) %.2f ms, idle %.2f ms in %d gaps" % ((iv[-1][1] - iv[0][0]) / 1e6, tot / 1e6, sum(g for g, _ in gaps) / 1e6, len(gaps)))
hist = collections.Counter()
for g, _ in gaps: hist[min(9, int(g / 1e4))] += g
print("idle by gap size (10us bins, last = >=90us):", " ".join("%.2f" % (hist[i] / 1e6) for i in range(10)))
if cp:
    c2 = [c for c in cp if c[0] >= t0]
    by = collections.Counter(); byb = collections.Counter()
    for s, e, dr, sz in c2: by[dr] += e - s; byb[dr] += sz
    for k in by: print("  copies %s: %.2f ms busy, %.1f MB" % (k, by[k] / 1e6, byb[k] / 1e6))
# how much of the backtrace (walk / emit / compact / wave-trace kernels) ran UNDER a wavefront kernel of another batch
def union(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def inter(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
al = union([(s, e) for s, e, k, _ in sel if "wfa_align_kernel" in k or "wfa_short" in k])
tr = union([(s, e) for s, e, k, _ in sel if any(t in k for t in ("wfa_walk", "wfa_emit", "wfa_text_compact", "wfa_trace"))])
tr_tot = sum(e - s for s, e in tr)
if tr_tot:
    print("backtrace kernels busy (union) %.2f ms, of which %.2f ms (%.0f %%) under a wavefront kernel of another batch" % (tr_tot / 1e6, inter(al, tr) / 1e6, 100.0 * inter(al, tr) / tr_tot))
