# which backtrace kernel differs between two runs of the same bench command?  (cfg3 trace: 3.31 ms in some processes, 3.49 in others)
R=$PWD; O=$R/gpurun_out/tracevar; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$i -o t -- python3 $R/bench.py --workload cfg3 --steps 6 --warmup 2 --no-cpu-baseline --no-host-to-host --no-configs > $O/log$i.txt 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$O/r$i/**/t_kernel_stats.csv',recursive=True)[0]
d={r['Name'].split('(')[0].split('::')[-1][:40]:float(r['AverageNs'])/1e6 for r in csv.DictReader(open(f))}
import json
line=open('$O/log$i.txt').read().strip().splitlines()[-1]
tr=json.loads(line)['stage_ms_per_step']['trace']
print('run $i trace %.3f | walk %.3f emit %.3f compact %.3f'%(tr, d.get('wfa_walk_kernel',0), [v for k,v in d.items() if 'wfa_emit_kernel' in k][0], d.get('wfa_text_compact_kernel',0)))
PY
done
