# round 6, GPU call 1: new tests + deterministic-rule A/B runs (capture branches, Cholesky stream sets)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6_run1; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed_gpu.py tests/test_nets.py -k "captured or two_ranks" -x -q > $O/t_capture.log 2>&1; echo "capture tests rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "cholesky or fails_soft" -x -q > $O/t_chol.log 2>&1; echo "chol tests rc=$?" >> $O/summary.txt
pick() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); k=d['kfac']
print('$1', 'ms_per_step %.4f' % d['ms_per_step'], 'kfac %.2f' % k['ms_per_batch'], 'route', k.get('route'), 'inv first %.1f second %.1f mean4 %.1f' % (k['cholesky_inverse_ms_first_call'], k['cholesky_inverse_ms_second_call'], k['cholesky_inverse_ms_mean_of_4']), 'c3', d.get('other_points',{}).get('c3_lenet5'))"; }
for q in "" "GPU_MAX_HW_QUEUES=4" ; do
for cfg in "CLO_KFAC_CAPTURE_BRANCHES=1 CLO_CHOL_OWN_MAIN=1" "CLO_KFAC_CAPTURE_BRANCHES=2 CLO_CHOL_OWN_MAIN=0" "CLO_KFAC_CAPTURE_BRANCHES=1 CLO_CHOL_PIPE=0"; do
  for rep in 1 2; do
  env $q $cfg timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_err.log | pick "[$q $cfg #$rep]" >> $O/ab.txt 2>&1
  done
done
done
CLO_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/ab.txt; tail -5 $O/t_capture.log $O/t_chol.log; tail -c 1500 $O/bench_gloo2.json
