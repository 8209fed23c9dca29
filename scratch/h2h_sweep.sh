# host to host, 1M x 1 kbp: lanes x batches per device (scratch/hostpath.py args 7, 8); median of the warm calls
for cfg in "0 0" "3 8" "3 12" "3 16" "3 24" "2 8" "2 16" "4 16" "4 24" "3 10"; do
  set -- $cfg
  echo "lanes=$1 batches=$2: $(python3 scratch/hostpath.py 1000000 1000 0.05 cigar 9 1000000 $1 $2 2>/dev/null | grep '^call' | awk '{print $3}' | tail -7 | sort -n | tr '\n' ' ')"
done
