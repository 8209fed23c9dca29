#!/bin/bash
# Collects the evidence for one workload on the GPU box: rocprofv3 kernel stats, PMC passes, bench line.
#   scratch/profile_round.sh <tag> <workload> [extra bench args...]      e.g.  scratch/profile_round.sh r02 cfg3
# Writes gpurun_out/prof_<tag>_<workload>/ and copies the summaries into profiles/<tag>/ (commit those).
TAG=${1:-r02}; WL=${2:-cfg3}; shift; shift
R=$PWD; O=$R/gpurun_out/prof_${TAG}_$WL; mkdir -p $O $R/profiles/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $WL -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-host-to-host --no-configs "$@" > $O/stats.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/$T -o pmc -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host --no-configs "$@" > $O/pmc_$T.log 2>&1
done
cd $R
python3 - <<PY
import csv,glob,re,sys
sys.path.insert(0,'$R')
import bench
rows=[]
for f in sorted(glob.glob('$O/pmc/*/pmc_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        if 'wfa_' in kn:
            k=re.search(r'wfa_\\w+(<[^>]*>)?',kn).group(0)
            rows.append((k,r['Counter_Name'],r['Counter_Value'],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,r['Grid_Size'],r['Workgroup_Size'],r['VGPR_Count'],r['SGPR_Count'],r.get('LDS_Block_Size','')))
with open('$R/profiles/$TAG/${WL}_pmc_counters.csv','w') as f:
    f.write('# rocprofv3 --pmc passes (one warm-up step + one step each; bench.py uses the largest launch per counter = the steady-state main launch) of: python3 bench.py --workload $WL --steps 1 --warmup 1 $@ ; kernel_sha1=%s\n' % bench.kernel_source_hash())
    w=csv.writer(f); w.writerow(['kernel','counter','value','kernel_ms_under_pmc','grid','wg','vgpr','sgpr','lds'])
    for r in rows: w.writerow(r)
print(len(rows),'pmc rows')
PY
cp $O/stats/*kernel_stats.csv $R/profiles/$TAG/${WL}_kernel_stats.csv 2>/dev/null
python3 bench.py --workload $WL --no-configs "$@" 2>$O/bench.err | tail -1 > $R/profiles/$TAG/bench_$WL.json
cp $R/profiles/$TAG/bench_$WL.json $R/profiles/$TAG/${WL}_pmc_counters.csv $R/profiles/$TAG/${WL}_kernel_stats.csv $O/ 2>/dev/null
cat $R/profiles/$TAG/bench_$WL.json; echo; head -8 $R/profiles/$TAG/${WL}_kernel_stats.csv
