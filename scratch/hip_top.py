"""Summary of a rocprofv3 --hip-trace CSV: per-function totals and the longest calls with their start times (ms since the
first HIP call of the process) and threads."""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((r["Function"], int(r["Thread_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if not rows:
    print("no hip api rows in", files); sys.exit(0)
t0 = min(r[2] for r in rows)
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for fn, tid, s, e in rows:
    t = tot[fn]; t[0] += 1; t[1] += (e - s) / 1e6; t[2] = max(t[2], (e - s) / 1e6)
print("function                               calls   total ms    max ms")
for fn, (n, ms, mx) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{fn:38s} {n:6d} {ms:10.2f} {mx:9.2f}")
tids = {t: i for i, t in enumerate(sorted({r[1] for r in rows}))}
print("\nlongest calls: start ms, duration ms, thread, function")
for fn, tid, s, e in sorted(rows, key=lambda r: r[2] - r[3])[:60]:
    print(f"{(s - t0) / 1e6:10.2f} {(e - s) / 1e6:9.2f}  t{tids[tid]:<3d} {fn}")
