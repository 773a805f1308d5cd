# round 6, GPU call 2: the 2.4 s stall of K.inverse; chain kernel stats at 16 / 32 / 64 rows
cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run2; mkdir -p $O
for rep in 1 2 3; do
  python tools/probe_kfac_inverse.py >> $O/inv_new.txt 2>&1
  CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_r05.so python tools/probe_kfac_inverse.py >> $O/inv_r05lib.txt 2>&1
done
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pinv -o k -- python $R/tools/probe_kfac_inverse.py > $O/inv_prof.log 2>&1
python $R/tools/r6/timeline.py /tmp/pinv/*/k_results.db > $O/inv_timeline.txt 2>&1 || python $R/tools/r6/timeline.py /tmp/pinv/k_results.db > $O/inv_timeline.txt 2>&1
for n in 16 32 64; do
rm -rf /tmp/pr$n
rocprofv3 --kernel-trace --stats -d /tmp/pr$n -o k -- python $R/tools/probe_c2.py $n > $O/log$n.txt 2>&1
db=$(ls /tmp/pr$n/*/k_results.db /tmp/pr$n/k_results.db 2>/dev/null | head -1)
python $R/tools/prof_summary.py $db $O/r06_c2_n${n}_kernel_stats_before.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_c2.py $n  (C2 GGN matvec, $n rows)"
head -16 $O/r06_c2_n${n}_kernel_stats_before.txt
done
cd $R
cat $O/inv_new.txt $O/inv_r05lib.txt $O/inv_timeline.txt
