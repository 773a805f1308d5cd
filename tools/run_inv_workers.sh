out=gpurun_out/invworkers; mkdir -p $out
for w in 1 2 4; do for rep in 1 2; do
  INV_WORKERS=$w python tools/bench_workers.py 12 > $out/b_${w}_$rep.json 2>/dev/null
  python - $out/b_${w}_$rep.json $w $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kfac"]; o = d["other_points"]
print(f"inverse workers {sys.argv[2]} run {sys.argv[3]}: cholesky inverse second call {k['cholesky_inverse_ms_second_call']:.2f} mean of 4 {k['cholesky_inverse_ms_mean_of_4']:.2f} | c3 inverse {o['c3_kfac_lenet5']['inverse_ms']:.2f} | eigh_ms {o['c4_ekfac_resnet18']['eigh_ms']:.1f}")
PY
done; done
