# round 6, GPU call 3: tall-skinny kernels + decaying-spectrum trace parity; the 2.4 s stall of K.inverse inside bench.py;
# inverse workers x pipeline in the bench process
cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run3; mkdir -p $O
timeout 900 python -m pytest tests/test_operators_gpu.py -k "trace or tall" -x -q > $O/t_trace.log 2>&1; echo "trace tests rc=$?" >> $O/summary.txt
tail -15 $O/t_trace.log
pick() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); k=d['kfac']
print('$1', 'kfac %.2f' % k['ms_per_batch'], 'inv first %.1f second %.1f mean4 %.1f' % (k['cholesky_inverse_ms_first_call'], k['cholesky_inverse_ms_second_call'], k['cholesky_inverse_ms_mean_of_4']), 'hutchpp', d.get('other_points',{}).get('c5_encoder'))"; }
for q in "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=4" ; do
for cfg in "CLO_CHOL_PIPE=0 CLO_INV_WORKERS=1" "CLO_CHOL_PIPE=0 CLO_INV_WORKERS=2" "CLO_CHOL_PIPE=0 CLO_INV_WORKERS=3" "CLO_CHOL_PIPE=1 CLO_INV_WORKERS=2"; do
  env $q $cfg timeout 600 python bench.py --steps 20 --warmup 5 2>>$O/bench_err.log | pick "[$q $cfg]" >> $O/ab.txt 2>&1
done
done
cat $O/ab.txt
cd /tmp
CLO_CHOL_PIPE=0 rocprofv3 --kernel-trace --stats -d /tmp/pb -o k -- python $R/bench.py --steps 20 --warmup 5 > $O/bench_prof.log 2>&1
db=$(ls /tmp/pb/*/k_results.db /tmp/pb/k_results.db 2>/dev/null | head -1)
python $R/tools/r6/timeline.py $db > $O/bench_timeline.txt 2>&1
tail -1 $O/bench_prof.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kfac']; print('profiled bench: inv first %.1f second %.1f' % (k['cholesky_inverse_ms_first_call'], k['cholesky_inverse_ms_second_call']))"
cat $O/bench_timeline.txt
grep -i "warn\|retry\|error" $O/bench_err.log | head
