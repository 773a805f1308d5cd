cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6_run16; mkdir -p $O
timeout 900 python -m pytest tests/test_nets.py tests/test_distributed_gpu.py -m gpu -q -k "captured or two_ranks" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2>$O/err.log > $O/bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_run16/bench.json') if l.startswith('{')][-1]); k=d['kfac']
print('ms_per_step %.4f' % d['ms_per_step'], 'stale', d['roofline'].get('traffic_stale'), 'kfac', k['ms_per_batch_median_min_max'], k['route'], 'roofline', k['roofline'].get('frac'), k['roofline']['clo_kernels']['source'][:60])
PY
CLO_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 > $O/gloo2.json 2> $O/gloo2.err; echo "gloo2 rc=$?"
