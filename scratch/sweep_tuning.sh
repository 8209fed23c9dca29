# scratch/sweep_tuning.sh <workload> <steps> <KEY> <v1> <v2> ...: one tuning field over several values, two rounds
WL=$1; ST=$2; K=$3; shift; shift; shift
for r in 1 2; do for v in "$@"; do
  python3 bench.py --workload $WL --steps $ST --warmup 3 --no-configs --no-cpu-baseline --no-host-to-host --tuning $K=$v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$K=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['tiers']['blocks_per_cu_first'], d['parity_sample']['bit_exact_vs_oracle'])"
done; done
