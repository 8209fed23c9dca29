# host to host, 1M x 1 kbp: every n-th batch as ASCII (wfagpu_amd_launch_config_t::ascii_every) against all batches packed on the host; alternating processes
for r in 1 2 3; do for v in -1 3 2 4; do
  echo "ascii_every=$v round $r: $(ASCII_EVERY=$v python3 scratch/hostpath.py 1000000 1000 0.05 cigar 9 2>/dev/null | grep '^call' | awk '{print $3}' | tr '\n' ' ')"
done; done
