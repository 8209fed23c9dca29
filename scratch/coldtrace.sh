#!/bin/bash
# scratch/coldtrace.sh: what a COLD launch_alignments call (every CLI invocation is one) spends its time on.
#   1. the CLI on 1M cfg3 pairs from a .seq file, twice (the "Wall time" line is the reference's metric, tools/aligner.c:450-474)
#   2. the same under rocprofv3 --hip-trace: the longest HIP API calls of the process with their start times
# -> gpurun_out/coldtrace/{cli.txt,hip_top.txt}
R=$PWD; O=$R/gpurun_out/coldtrace; rm -rf $O; mkdir -p $O
SEQ=/tmp/cfg3_cold.seq
$R/wfa-gpu_amd/bin/generate_dataset -n ${1:-1000000} -l 1000 -e 0.05 -s 9 -t 16 -o $SEQ
for i in 1 2; do
  $R/wfa-gpu_amd/bin/wfa.affine.gpu -i $SEQ -x -e 300 -o /tmp/cfg3_cold.out 2>&1 | grep -v "^\[Info\] Using" >> $O/cli.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --output-format csv -d $O/tr -o cold -- $R/wfa-gpu_amd/bin/wfa.affine.gpu -i $SEQ -x -e 300 > $O/cli_traced.txt 2>&1
cd $R
python3 scratch/hip_top.py $O/tr > $O/hip_top.txt 2>&1
rm -rf $O/tr
cat $O/cli.txt $O/cli_traced.txt; head -80 $O/hip_top.txt
