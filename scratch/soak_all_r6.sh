mkdir -p gpurun_out/r6soak
O=gpurun_out/r6soak/soak.txt; : > $O
for s in 61 62 63; do timeout 300 python3 scratch/soak.py $s 100 2>&1 | tail -1 >> $O; done
for s in 71 72; do timeout 300 python3 scratch/soak_autobudget.py $s 12 2>&1 | tail -1 >> $O; done
for s in 81 82 83 84 85 86; do timeout 300 python3 scratch/soak_banded.py $s 40 2>&1 | tail -2 >> $O; done
for s in 91 92 93; do timeout 300 python3 scratch/soak_launch.py $s 30 2>&1 | tail -1 >> $O; done
for s in 101 102 103 104 105 106; do timeout 300 python3 scratch/soak_short.py $s 80 2>&1 | tail -1 >> $O; done
for s in 111 112 113 114; do timeout 400 python3 scratch/soak_long.py $s 25 2>&1 | tail -1 >> $O; done
cat $O
