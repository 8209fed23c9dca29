# does the generator's dirty page cache slow the CLI run that follows it?  gen -> CLI, CLI | gen -> sync -> CLI, CLI
T=$(mktemp -d)
run() { ./wfa-gpu_amd/bin/wfa.affine.gpu -i $T/a.seq -x -e 300 --stage-times 2>&1 | grep "Wall time\|wfagpu timing\] device" | sed 's/.*acquire/acquire/' | tr '\n' ' '; echo; }
grep -i "dirty\|writeback" /proc/meminfo | tr '\n' ' '; echo
./wfa-gpu_amd/bin/generate_dataset -n 1000000 -l 1000 -e 0.05 -s 9 -t 16 -o $T/a.seq
grep -i "dirty\|writeback" /proc/meminfo | tr '\n' ' '; echo
echo "gen -> CLI"; run; run
rm $T/a.seq
./wfa-gpu_amd/bin/generate_dataset -n 1000000 -l 1000 -e 0.05 -s 9 -t 16 -o $T/a.seq
/usr/bin/time -f "sync %e s" sync
grep -i "dirty\|writeback" /proc/meminfo | tr '\n' ' '; echo
echo "gen -> sync -> CLI"; run; run
rm -rf $T
