# band_probe.py's ONT grid with several builds of the library, alternating:  scratch/ab_probe.sh <name> <name> ...
cp wfa-gpu_amd/libwfagpu.so /tmp/lib_keep.so
for i in 1 2; do for v in "$@"; do
  cp scratch/lib_$v.so wfa-gpu_amd/libwfagpu.so
  echo "== $v"; python3 scratch/band_probe.py ont 1024 0 2>/dev/null | grep -E "lambda (25|100):" | awk '{print $2,$3,$4,$5,"main",$10}'
done; done
cp /tmp/lib_keep.so wfa-gpu_amd/libwfagpu.so
