cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run7; mkdir -p $O
for rep in 1 2; do
python tools/r6/probe_syrk_grouped.py 2>&1 | grep -v amdgpu | tee -a $O/sg.txt
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_sgv1.so python tools/r6/probe_syrk_grouped.py 2>&1 | grep -v amdgpu | tee -a $O/sg.txt
done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_distributed_gpu.py -m gpu -q -k "syrk_grouped or two_ranks" > $O/t_sel.log 2>&1; echo "selected tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_sel.log
CLO_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r6_run7/bench_gloo2.json') if l.startswith('{')][-1])
print(d['n_gt1']); print(d['kfac']['route'], d['kfac'].get('parts'), d['kfac']['ms_per_batch'])
PY
