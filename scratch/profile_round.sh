#!/bin/bash
# Collects the round's evidence on the GPU box: bench lines, rocprofv3 kernel stats, PMC passes.
TAG=${1:-r01}
R=$PWD; O=$R/gpurun_out/final_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o cfg3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/$T -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$T.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob,re
rows=[]
for f in sorted(glob.glob('$O/pmc/*/pmc_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        if 'wfa_' in kn:
            k=re.search(r'wfa_\\w+(<[^>]*>)?',kn).group(0)
            rows.append((k,r['Counter_Name'],r['Counter_Value'],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,r['Grid_Size'],r['Workgroup_Size'],r['VGPR_Count'],r['SGPR_Count']))
with open('$O/pmc_counters.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','counter','value','kernel_ms_under_pmc','grid','wg','vgpr','sgpr'])
    for r in rows: w.writerow(r)
print(len(rows),'pmc rows')
PY
mkdir -p $R/profiles/$TAG && cp $O/pmc_counters.csv $R/profiles/$TAG/cfg3_pmc_counters.csv   # bench.py reads traffic / issue counts from it
cp $O/pmc_counters.csv $O/cfg3_pmc_counters.csv
python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_cfg3.json
python bench.py --workload cfg2 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_cfg2.json
cat $O/bench_cfg3.json; echo; cat $O/bench_cfg2.json; echo; head -8 $O/stats/*kernel_stats.csv
