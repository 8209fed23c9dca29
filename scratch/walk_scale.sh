# walk kernel time per pair against the batch size (is the backtrace's random access bound by the footprint -- TLB reach -- or by HBM?)
R=$PWD; O=$R/gpurun_out/walkscale; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 125000 250000 500000 1000000 2000000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$n -o t -- python3 $R/bench.py --workload cfg3 --pairs $n --steps 4 --warmup 2 --no-cpu-baseline --no-host-to-host --no-configs > $O/log$n.txt 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$O/r$n/**/t_kernel_stats.csv',recursive=True)[0]
d={}
for r in csv.DictReader(open(f)): d[r['Name']]=float(r['AverageNs'])/1e6
w=[v for k,v in d.items() if 'wfa_walk_kernel' in k][0]; e=[v for k,v in d.items() if 'wfa_emit_kernel' in k][0]; c=[v for k,v in d.items() if 'wfa_text_compact' in k][0]
a=max(v for k,v in d.items() if 'wfa_align_kernel' in k)
print('pairs %8d: walk %.3f ms (%.2f ns/pair) emit %.3f (%.2f) compact %.3f | align main %.2f (%.2f ns/pair)'%($n, w, w*1e6/$n, e, e*1e6/$n, c, a, a*1e6/$n))
PY
done
