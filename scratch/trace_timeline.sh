#!/bin/bash
# kernels of the backtrace of the last configs[2] step on one clock (start, duration):  scratch/trace_timeline.sh [bench args]
R=$PWD; O=$R/gpurun_out/tracetl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o st -- python3 $R/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-host-to-host --no-configs "$@" > $O/log.txt 2>&1
cd $R
python3 - <<PY
import csv,glob,re
ker=[]
for f in glob.glob('$O/tr/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): ker.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])[:40], r.get('Queue_Id','')))
ker.sort()
walks=[k for k in ker if 'walk' in k[2]]
# last step: kernels from the last main align kernel on
big=[k for k in ker if 'wfa_align_kernel' in k[2] and k[1]-k[0] > 5e6]
t0=big[-1][0]
for k in ker:
    if k[0] >= t0: print(f"{(k[0]-t0)/1e3:10.1f} us  {(k[1]-k[0])/1e3:9.1f} us  q{k[3]}  {k[2]}")
PY
rm -rf $O/tr
