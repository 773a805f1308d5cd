R=$PWD; OUT=$R/gpurun_out/r05_rows; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for n in 128 65; do
rm -rf /tmp/pr$n
rocprofv3 --kernel-trace --stats -d /tmp/pr$n -o k -- python $R/tools/probe_c2.py $n > $OUT/log$n.txt 2>&1
python $R/tools/prof_summary.py /tmp/pr$n/k_results.db $OUT/r05_c2_n${n}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_c2.py $n  (C2 GGN matvec, $n rows, kernel path; 35 products)"
cat $OUT/r05_c2_n${n}_kernel_stats.txt | head -24
done
