# the CLI's own stage times on a 1M x 1 kbp file, three fresh processes
T=$(mktemp -d); ./wfa-gpu_amd/bin/generate_dataset -n 1000000 -l 1000 -e 0.05 -s 9 -t 16 -o $T/a.seq
for i in 1 2 3; do /usr/bin/time -f "process %e s (user %U sys %S)" ./wfa-gpu_amd/bin/wfa.affine.gpu -i $T/a.seq -x -e 300 --stage-times 2>&1 | grep "cli stages\|process\|Wall"; done
rm -rf $T
