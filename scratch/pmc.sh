#!/bin/bash
# usage: scratch/pmc.sh <tag> [bench args...]   (run on the GPU box from the repo root)
TAG=$1; shift
R=$PWD; mkdir -p $R/gpurun_out/pmc_$TAG; cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_LDS"; do
  T=$(echo $C | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$T -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$TAG/$T.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob
for f in sorted(glob.glob('gpurun_out/pmc_$TAG/*/pmc_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'wfa_align' in r['Kernel_Name']:
            print(r['Counter_Name'], r['Counter_Value'], 'ms', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, 'vgpr', r['VGPR_Count'], 'sgpr', r['SGPR_Count'])
PY
