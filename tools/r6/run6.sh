# round 6, GPU call 6: failing tests again + distributed; grouped SYRK efficiency; bench at the runtime's default queues
cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_operators_gpu.py tests/test_distributed_gpu.py tests/test_nets.py -m gpu -q -k "syrk_grouped or trace or tall or gram_orthonormal or two_ranks or captured or fold or kfac" > $O/t_sel.log 2>&1; echo "selected tests rc=$?" >> $O/summary.txt
tail -8 $O/t_sel.log
cd /tmp; export MIOPEN_FIND_MODE=FAST
rm -rf /tmp/pkb
rocprofv3 --kernel-trace -d /tmp/pkb -o k -- python $R/tools/prof_kfac_build.py > /dev/null 2>&1
db=$(ls /tmp/pkb/*/k_results.db /tmp/pkb/k_results.db 2>/dev/null | head -1)
python $R/tools/kfac_trace_summary.py $db 512 > $O/kfac_build_kernels.txt 2>&1
head -16 $O/kfac_build_kernels.txt
unset MIOPEN_FIND_MODE; cd $R
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 2>>$O/bench_err.log > $O/bench_$rep.json
python -c "
import json
d=json.loads([l for l in open('$O/bench_$rep.json') if l.startswith('{')][-1]); k=d['kfac']; o=d.get('other_points',{})
print('bench $rep', 'ms_per_step %.4f' % d['ms_per_step'], 'kfac', k['ms_per_batch_median_min_max'], k['route'], 'inv', k['cholesky_inverse_ms_first_call'], k['cholesky_inverse_ms_median_min_max'], 'c3', o.get('c3_kfac_lenet5',{}).get('factor_build_ms_mc'), o.get('c3_kfac_lenet5',{}).get('factor_build_ms_type-2'), 'hutchpp', o.get('c5_encoder_ef_hutchpp',{}).get('hutchpp_96_ms'), 'roofline', k['roofline'].get('frac'))" | tee -a $O/summary.txt
done
CLO_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?" >> $O/summary.txt
tail -c 600 $O/bench_gloo2.json
