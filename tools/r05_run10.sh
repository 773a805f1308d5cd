R=$PWD; OUT=$R/gpurun_out/r05_run10; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for m in plain finecap; do
rm -rf /tmp/pi_$m
MARK=1 rocprofv3 --kernel-trace -d /tmp/pi_$m -o k -- python $R/tools/probe_inverse_bisect.py $m > $OUT/log_$m.txt 2>&1
grep "inverse calls" $OUT/log_$m.txt
python $R/tools/trace_section_queues.py /tmp/pi_$m/k_results.db > $OUT/queues_$m.txt 2>&1; cat $OUT/queues_$m.txt
done
