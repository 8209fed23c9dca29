# CIGAR replay: alignments per wavefront (tuning.emit_pairs) on configs[2]
for r in 1 2; do for e in 0 24 32 40 48 56 64; do
  python3 bench.py --workload cfg3 --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-host-to-host --tuning emit_pairs=$e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 emit_pairs', $e, d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done; done
