# the Cholesky-inverse line of bench.py's kfac leg under different settings (whole bench process each time)
for cfg in "CLO_CHOL_PIPE=0" "CLO_CHOL_PIPE=1" "CLO_CHOL_PIPE=1 CLO_INV_STREAMS=1" "CLO_CHOL_PIPE=1 CLO_CHOL_HELPER_PRIO=0" "CLO_CHOL_PIPE=1 CLO_INV_STREAMS=4" "CLO_CHOL_PIPE=0 CLO_INV_STREAMS=4"; do
  env $cfg python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kfac']
print('$cfg', 'inverse first %.1f second %.1f mean4 %.1f ms' % (k['cholesky_inverse_ms_first_call'], k['cholesky_inverse_ms_second_call'], k['cholesky_inverse_ms_mean_of_4']))"
done
