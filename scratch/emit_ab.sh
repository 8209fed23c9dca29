for e in 0 64 56 48 40 32 24; do
  python3 bench.py --workload cfg3 --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-host-to-host --tuning emit_pairs=$e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 emit_pairs', $e, d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['parity_sample'])"
done
for e in 0 64 48 32; do
  python3 bench.py --workload cfg2c --no-configs --no-cpu-baseline --no-host-to-host --tuning emit_pairs=$e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2c emit_pairs', $e, d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['parity_sample'])"
done
