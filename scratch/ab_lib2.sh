# scratch/ab_lib2.sh <workload> <steps> [extra bench args]: scratch/lib_old.so against scratch/lib_new.so, alternating
WL=$1; ST=$2; shift; shift
for i in 1 2 3; do for v in old new; do
  cp scratch/lib_$v.so wfa-gpu_amd/libwfagpu.so
  python3 bench.py --workload $WL --steps $ST --warmup 3 --no-configs --no-cpu-baseline --no-host-to-host "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel_ms'], d['tiers']['blocks_per_cu_first'], d['parity_sample']['bit_exact_vs_oracle'])"
done; done
cp scratch/lib_new.so wfa-gpu_amd/libwfagpu.so
