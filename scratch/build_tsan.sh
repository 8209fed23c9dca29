#!/bin/bash
# Builds tests/launch_tsan.cpp + csrc/wfa_launch.hip (as C++, stub HIP layer) under ThreadSanitizer into $1 (default /tmp/tsanb).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=${1:-/tmp/tsanb}; mkdir -p $O; cd $O; rm -f *.o
CXX="g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -I$R/tests/hip_stub -I$R/include"
$CXX -x c++ -c $R/wfa-gpu_amd/csrc/wfa_launch.hip -o launch.o
$CXX -c $R/tests/hip_stub/stub.cpp -o stub.o
$CXX -c $R/tests/launch_tsan.cpp -o harness.o
for f in oracle/wfa_oracle.c wfa-gpu_amd/lib/alignment_results.c wfa-gpu_amd/utils/verification.c wfa-gpu_amd/utils/host_pack.c wfa-gpu_amd/tools/generate_dataset.c; do
  gcc -O1 -g -fsanitize=thread -Wno-unknown-pragmas -I$R/include -c $R/$f -o $(basename $f .c).o      # (no -fopenmp: libgomp is not instrumented)
done
g++ -fsanitize=thread *.o -lpthread -lm -o launch_tsan
