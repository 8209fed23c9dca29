"""Lists the kernels of the last <window_ms> of a rocprofv3 --kernel-trace CSV directory: start, end, duration, queue, name.
  python scratch/klist.py <dir> [window_ms] [min_us]"""
import csv, glob, re, sys
d = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0; min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]; m = re.search(r"(wfa_\w+|k_\w+|__amd_rocclr_\w+)", n)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else n[:30], r.get("Queue_Id"), r.get("Grid_Size_X"), r.get("LDS_Block_Size")))
rows.sort()
t0 = rows[-1][1] - int(win * 1e6)
for s, e, k, q, g, lds in rows:
    if s >= t0 and (e - s) >= min_us * 1e3:
        print("%8.3f %8.3f %7.3f q%s %-26s grid %s lds %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, k, g, lds))
