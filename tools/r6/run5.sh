# round 6, GPU call 5: full GPU suite; KFAC build kernel trace with the grouped gradient covariances; capture branches x queues;
# nt weight loads in the persistent kernel (A/B)
cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run5; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "gpu suite rc=$?" >> $O/summary.txt
tail -12 $O/t_all.log
cd /tmp; export MIOPEN_FIND_MODE=FAST
rm -rf /tmp/pkb
rocprofv3 --kernel-trace -d /tmp/pkb -o k -- python $R/tools/prof_kfac_build.py > /dev/null 2>&1
db=$(ls /tmp/pkb/*/k_results.db /tmp/pkb/k_results.db 2>/dev/null | head -1)
python $R/tools/kfac_trace_summary.py $db 512 > $O/kfac_build_kernels.txt 2>&1
head -40 $O/kfac_build_kernels.txt
unset MIOPEN_FIND_MODE; cd $R
pick() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); k=d['kfac']; o=d.get('other_points',{})
print('$1', 'ms_per_step %.4f' % d['ms_per_step'], 'kfac %.2f' % k['ms_per_batch'], 'inv %.1f/%.1f' % (k['cholesky_inverse_ms_first_call'], k['cholesky_inverse_ms_second_call']), 'c3', o.get('c3_kfac_lenet5',{}).get('factor_build_ms_mc'), o.get('c3_kfac_lenet5',{}).get('factor_build_ms_type-2'), 'hutchpp', o.get('c5_encoder_ef_hutchpp',{}).get('hutchpp_96_ms'), 'eigh', {kk: vv for kk, vv in o.items() if 'ekfac' in kk})"; }
for cfg in "GPU_MAX_HW_QUEUES=16 CLO_KFAC_CAPTURE_BRANCHES=1" "GPU_MAX_HW_QUEUES=4 CLO_KFAC_CAPTURE_BRANCHES=2" "GPU_MAX_HW_QUEUES=4 CLO_KFAC_CAPTURE_BRANCHES=2" "GPU_MAX_HW_QUEUES=4 CLO_KFAC_CAPTURE_BRANCHES=1"; do
  env $cfg timeout 900 python bench.py --steps 20 --warmup 5 2>>$O/bench_err.log | pick "[$cfg]" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
for rep in 1 2 3; do
  python bench.py --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default  ms_per_step %.5f' % d['ms_per_step'])" >> $O/ntw.txt
  CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_ntw.so python bench.py --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt-loads ms_per_step %.5f' % d['ms_per_step'])" >> $O/ntw.txt
done
cat $O/ntw.txt
