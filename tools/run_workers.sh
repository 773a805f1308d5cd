out=gpurun_out/workers; mkdir -p $out
for w in 6 8 10 12; do for rep in 1 2; do
  python tools/bench_workers.py $w > $out/b_${w}_$rep.json 2>/dev/null
  python - $out/b_${w}_$rep.json $w $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kfac"]; o = d["other_points"]
print(f"workers {sys.argv[2]:>2s} run {sys.argv[3]}: eigh_ms {o['c4_ekfac_resnet18']['eigh_ms']:.1f} ekfac_total_ms {o['c4_ekfac_resnet18']['ekfac_total_ms']:.1f} | cholesky inverse second call {k['cholesky_inverse_ms_second_call']:.2f} mean of 4 {k['cholesky_inverse_ms_mean_of_4']:.2f} | kfac build {k['ms_per_batch']:.2f} | C2 {d['ms_per_step']*1e3:.1f} us")
PY
done; done
