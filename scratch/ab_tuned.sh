# scratch/ab_tuned.sh <workload> <steps> "<bench args>" <name> <name> ...: builds scratch/lib_<name>.so alternating on one box, with extra bench arguments
WL=$1; ST=$2; EXTRA=$3; shift; shift; shift
cp wfa-gpu_amd/libwfagpu.so /tmp/lib_keep.so
for i in 1 2 3; do for v in "$@"; do
  cp scratch/lib_$v.so wfa-gpu_amd/libwfagpu.so
  python3 bench.py --workload $WL --steps $ST --warmup 2 --no-configs --no-cpu-baseline --no-host-to-host $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL $v', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done; done
cp /tmp/lib_keep.so wfa-gpu_amd/libwfagpu.so
