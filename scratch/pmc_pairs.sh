#!/bin/bash
# Instruction counts of the main launch with and without two scores per chunk (tuning.no_score_pairs):  scratch/pmc_pairs.sh <workload>
WL=${1:-cfg3_x3o1e4}; R=$PWD
cd /tmp && export TMPDIR=/tmp
for T in 0 1; do
  O=$R/gpurun_out/pmcp_$T; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O -o pmc -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host --no-configs --tuning no_score_pairs=$T > $O/log.txt 2>&1
  python3 - <<PY
import csv,glob
best={}
for f in glob.glob('$O/*/pmc_counter_collection.csv')+glob.glob('$O/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'wfa_align' in r['Kernel_Name']:
            ms=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
            k=r['Counter_Name']
            if k not in best or ms>best[k][1]: best[k]=(float(r['Counter_Value']),ms)
print('no_score_pairs=$T', {k:('%.4g'%v[0], '%.2f ms'%v[1]) for k,v in sorted(best.items())})
PY
done
