# per-kernel average times of a short bench run:  scratch/kstats.sh [workload]
WL=${1:-cfg3}; R=$PWD; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -o t -- python3 $R/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-host-to-host --no-configs > $O/log.txt 2>&1
cd $R
python3 - <<PY
import csv,glob,re
f=glob.glob('$O/r/**/t_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print('%-70s calls %4s avg %9.3f ms  %5s %%' % (re.sub(r'\(anonymous namespace\)::','',r['Name'])[:70], r['Calls'], float(r['AverageNs'])/1e6, r['Percentage']))
PY
