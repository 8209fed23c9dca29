# long reads: wave kernel walks + lane kernel replays (default) against the wave kernel doing all of it (trace_mode=2)
set -x
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long or 10k or cigar or wave or trace" 2>&1 | tail -5
for r in 1 2; do for w in cfg4 cfg5; do for t in "trace_mode=0" "trace_mode=2" "emit_pairs=8"; do
  python3 bench.py --workload $w --steps 6 --warmup 2 --no-configs --no-cpu-baseline --no-host-to-host --tuning $t 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $t', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done; done; done
