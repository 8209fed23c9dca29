# the whole evidence pass of round 6 on one box:  scratch/profile_all_r6.sh
TAG=r06
for w in cfg3 cfg3_x5o3e2 cfg3_x3o1e4 cfg2 cfg2c cfg2_x5o3e2 cfg4 cfg4b cfg4x cfg5 cfg5t3; do bash scratch/profile_round.sh $TAG $w > gpurun_out/profile_$w.log 2>&1; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default_driverlike.json
python3 bench.py --ont-banded-only 2>/dev/null | tail -1 > gpurun_out/ont_banded_grid.json
