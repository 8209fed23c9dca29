# scratch/ab_h2h.sh <workload>: host-to-host calls of scratch/lib_old.so against scratch/lib_new.so, alternating
WL=${1:-cfg3}
for i in 1 2 3; do for v in old new; do
  cp scratch/lib_$v.so wfa-gpu_amd/libwfagpu.so
  python3 bench.py --workload $WL --steps 2 --warmup 1 --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_to_host']; s=h['stages_ms']; print('$v', h['pageable']['warm_ms'], h['pageable']['best_ms'], h['pageable']['calls_ms'][3:], 'prep', s['prep_ms'], 'pack', s['host_pack_ms'], 'scatter', s['scatter_ms'])"
done; done
cp scratch/lib_new.so wfa-gpu_amd/libwfagpu.so
