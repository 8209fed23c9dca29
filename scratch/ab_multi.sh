# several builds of the library on one box, alternating:  scratch/ab_multi.sh <workload> <steps> <name> <name> ...   (scratch/lib_<name>.so)
WL=$1; ST=$2; shift; shift
cp wfa-gpu_amd/libwfagpu.so /tmp/lib_keep.so
for i in 1 2 3; do for v in "$@"; do
  cp scratch/lib_$v.so wfa-gpu_amd/libwfagpu.so
  python3 bench.py --workload $WL --steps $ST --warmup 3 --no-configs --no-cpu-baseline --no-host-to-host 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel_ms'], d['parity_sample'])"
done; done
cp /tmp/lib_keep.so wfa-gpu_amd/libwfagpu.so
