# scratch/sweep_tuning_pairs.sh <workload> <pairs> <steps> <KEY> <v1> <v2> ...
WL=$1; NP=$2; ST=$3; K=$4; shift; shift; shift; shift
for r in 1 2; do for v in "$@"; do
  python3 bench.py --workload $WL --pairs $NP --steps $ST --warmup 3 --no-configs --no-cpu-baseline --no-host-to-host --tuning $K=$v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NP $K=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['tiers']['blocks_per_cu_first'], d['parity_sample']['bit_exact_vs_oracle'])"
done; done
