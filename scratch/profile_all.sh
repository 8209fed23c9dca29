# the whole evidence pass of a round on one box:  scratch/profile_all.sh <tag>
TAG=${1:-r04}
for w in cfg3 cfg2 cfg2c cfg4 cfg4b cfg5; do bash scratch/profile_round.sh $TAG $w > gpurun_out/profile_$w.log 2>&1; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default_driverlike.json
