# scratch/ab_pairs.sh <workload> <pairs> <steps> <KEY=INT>: alternating runs with and without a tuning switch at a batch size
WL=$1; NP=$2; ST=$3; KV=$4
for i in 1 2 3; do for v in base "$KV"; do
  if [ "$v" = base ]; then T=""; else T="--tuning $v"; fi
  python3 bench.py --workload $WL --pairs $NP --steps $ST --warmup 3 --no-configs --no-cpu-baseline --no-host-to-host $T 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel_ms'], d['tiers']['blocks_per_cu_first'], d['parity_sample'])"
done; done
