R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r24
cd /tmp && export TMPDIR=/tmp
for n in 16 32 64; do
rm -rf /tmp/pr$n
rocprofv3 --kernel-trace --stats -d /tmp/pr$n -o k -- python $R/tools/probe_c2.py $n > /tmp/probe_$n.log 2>&1
grep "N=" /tmp/probe_$n.log
python $R/tools/prof_summary.py /tmp/pr$n/k_results.db $R/gpurun_out/r24/c2_n${n}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_c2.py $n"
cut -c1-130 $R/gpurun_out/r24/c2_n${n}_kernel_stats.txt | head -10 | tail -8
done
