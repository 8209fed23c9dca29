"""One step of a bench run under rocprofv3 --hip-trace --kernel-trace: every HIP call and every kernel of the LAST `n` steps'
window, on one time axis (us).  usage: step_timeline.py <dir> [window_us_from_end]"""
import csv, glob, sys
d = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
ev = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERNEL", r["Kernel_Name"][:60]))
ev.sort()
# the last align kernel of the run marks the end of the window
last = max(e[1] for e in ev if e[2] == "KERNEL" and "wfa_short" in e[3])
t0 = last - win * 1000
for s, e, k, n in ev:
    if s >= t0 and s <= last + 60000:
        print(f"{(s - t0) / 1000:9.2f} {(e - s) / 1000:8.2f}  {k:6s} {n}")
