# tier 5: resident wavefronts per CU against time (100k and 1M short pairs, score-only and with CIGARs)
for wl in cfg2 cfg2c; do for p in 100000 1000000; do for b in 0 20 14 10; do
  python3 bench.py --workload $wl --pairs $p --steps 100 --warmup 3 --no-configs --no-cpu-baseline --no-host-to-host --tuning max_blocks_per_cu=$b 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl pairs', $p, 'max_blocks_per_cu', $b, d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
