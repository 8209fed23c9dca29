# more seeds of every soak (scratch/soak_all.sh's programs): gpurun_out/r6soak/soak_more.txt
mkdir -p gpurun_out/r6soak
O=gpurun_out/r6soak/soak_more.txt; : > $O
for s in 161 162 163 164 165 166; do timeout 300 python3 scratch/soak.py $s 100 2>&1 | tail -1 >> $O; done
for s in 171 172 173; do timeout 300 python3 scratch/soak_autobudget.py $s 12 2>&1 | tail -1 >> $O; done
for s in 181 182 183 184 185 186 187 188; do timeout 300 python3 scratch/soak_banded.py $s 40 2>&1 | tail -1 >> $O; done
for s in 191 192 193 194 195 196; do timeout 300 python3 scratch/soak_launch.py $s 30 2>&1 | tail -1 >> $O; done
for s in 201 202 203 204 205 206 207 208 209 210 211 212; do timeout 300 python3 scratch/soak_short.py $s 80 2>&1 | tail -1 >> $O; done
for s in 221 222 223 224; do timeout 400 python3 scratch/soak_long.py $s 25 2>&1 | tail -1 >> $O; done
cat $O
