#!/bin/bash
# scratch/cold_long.sh: where do the first calls of a long-read launch_alignments go?  cfg5 (1024 x 30 kbp @ 10 %), three calls of
# a fresh process, stage clocks + HIP API trace + kernel trace -> gpurun_out/cold_long/
R=$PWD; O=$R/gpurun_out/cold_long; rm -rf $O; mkdir -p $O
TIMING=2 python3 scratch/hostpath.py 1024 30000 0.10 cigar 3 > $O/plain.txt 2>&1
cd /tmp && export TMPDIR=/tmp
TIMING=1 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/tr -o cold -- python3 $R/scratch/hostpath.py 1024 30000 0.10 cigar 2 > $O/traced.txt 2>&1
cd $R
python3 scratch/hip_top.py $O/tr > $O/hip_top.txt 2>&1
python3 scratch/kernel_top.py $O/tr 40 > $O/kernel_top.txt 2>&1
rm -rf $O/tr
grep -v "batch" $O/plain.txt | head -30; head -40 $O/hip_top.txt; head -45 $O/kernel_top.txt
