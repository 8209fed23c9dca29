#!/bin/bash
# Two PMC passes over the tier-5 launch of a short-read workload.   scratch/pmc_short.sh [cfg2|cfg2c]
WL=${1:-cfg2c}; R=$PWD; O=$R/gpurun_out/pmcs_$WL; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INSTS_FLAT"; do
  T=$(echo $C | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -o pmc -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host --no-configs > $O/log_$T.txt 2>&1
done
cd $R
python3 - <<PY
import csv,glob
best={}
for f in glob.glob('$O/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'wfa_short' in r['Kernel_Name']:
            ms=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
            k=r['Counter_Name']
            if k not in best or ms>best[k][1]: best[k]=(float(r['Counter_Value']),ms,r['VGPR_Count'],r['SGPR_Count'],r.get('LDS_Block_Size',''))
for k,v in sorted(best.items()): print('%-24s %14.5g  (launch %.3f ms, vgpr %s sgpr %s lds %s)'%(k,v[0],v[1],v[2],v[3],v[4]))
PY
