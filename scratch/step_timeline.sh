#!/bin/bash
# One resident step of a workload as the host and the GPU see it: HIP API calls (start, duration) and kernels (start, duration, gap)
# of the LAST step of a short bench run, on one clock.   scratch/step_timeline.sh [workload]   -> gpurun_out/steptl_<workload>.txt
WL=${1:-cfg2}; R=$PWD; O=$R/gpurun_out/steptl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/tr -o st -- python3 $R/bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-host-to-host --no-configs > $O/log.txt 2>&1
cd $R
python3 - > gpurun_out/steptl_$WL.txt <<PY
import csv,glob,re
api=[]; ker=[]
for f in glob.glob('$O/tr/**/*hip_api_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): api.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Function']))
for f in glob.glob('$O/tr/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): ker.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])[:50]))
api.sort(); ker.sort()
packs=[k for k in ker if 'wfa_pack_kernel' in k[2]]
t0=packs[-1][0]
# the API calls of the last step: from the launch that precedes the last pack kernel
start=max(a[0] for a in api if a[0] < t0 and a[2].startswith('hipLaunchKernel') or a[0] < t0 and a[2].startswith('hipExtLaunch')) if api else t0
ev=[(a[0],'A',a) for a in api if a[0]>=start-50000]+[(k[0],'K',k) for k in ker if k[0]>=t0]
ev.sort()
pe=None
for t,kind,x in ev:
    if kind=='A': print(f"{(t-t0)/1e3:9.1f} us  host  {(x[1]-x[0])/1e3:7.1f} us  {x[2]}")
    else:
        gap=(x[0]-pe)/1e3 if pe else 0.0
        print(f"{(t-t0)/1e3:9.1f} us  GPU   {(x[1]-x[0])/1e3:7.1f} us  gap {gap:6.1f}  {x[2]}"); pe=x[1]
PY
rm -rf $O/tr
tail -80 gpurun_out/steptl_$WL.txt
