"""Longest kernel dispatches of a rocprofv3 --kernel-trace CSV with their start times (ms since the first dispatch)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"][:90], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if not rows:
    print("no kernel rows"); sys.exit(0)
t0 = min(r[1] for r in rows)
print("start ms, duration ms, kernel")
for name, s, e in sorted(rows, key=lambda r: r[1] - r[2])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{(s - t0) / 1e6:10.2f} {(e - s) / 1e6:9.3f}  {name}")
