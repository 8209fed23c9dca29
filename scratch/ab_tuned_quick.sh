# alternating runs of one library with and without a tuning switch:  scratch/ab_tuned_quick.sh <workload> <steps> <KEY=INT>
WL=$1; ST=$2; KV=$3
for i in 1 2 3; do for v in base "$KV"; do
  if [ "$v" = base ]; then T=""; else T="--tuning $v"; fi
  python3 bench.py --workload $WL --steps $ST --warmup 3 --no-configs --no-cpu-baseline --no-host-to-host $T 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel_ms'], d['tiers']['pairs_per_tier'], d['tiers']['blocks_per_cu_first'], d['parity_sample'])"
done; done
