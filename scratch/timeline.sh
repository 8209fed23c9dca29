#!/bin/bash
# scratch/timeline.sh <lanes> <batches>: kernel timeline of warm launch_alignments calls (cfg3) -> gpurun_out/tl_<lanes>_<batches>.txt
L=${1:-1}; B=${2:-16}; R=$PWD; O=$R/gpurun_out/tl_${L}_$B; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o tl -- python3 $R/scratch/hostpath.py 1000000 1000 0.05 cigar 4 1000000 $L $B > $O/run.log 2>&1
cd $R
grep "^call" $O/run.log
python3 scratch/timeline.py $O 55 > $R/gpurun_out/tl_${L}_$B.txt 2>&1
cat $R/gpurun_out/tl_${L}_$B.txt
rm -rf $O/*/*.csv 2>/dev/null; find $O -name "*.csv" -size +20M -delete
