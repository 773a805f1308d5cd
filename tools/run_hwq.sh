# does the number of HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) move the multi-stream legs of the bench line?
out=gpurun_out/hwq3; mkdir -p $out
for q in 16 24 32; do
  for rep in 1 2; do
    GPU_MAX_HW_QUEUES=$q python bench.py > $out/bench_q${q}_$rep.json 2> /dev/null
    python - $out/bench_q${q}_$rep.json $q $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kfac"]; o = d["other_points"]
print(f"queues {sys.argv[2]} rep {sys.argv[3]}: ms_per_step {d['ms_per_step']*1e3:.1f} us | kfac build {k['ms_per_batch']:.2f} ms | chol inverse 2nd {k.get('cholesky_inverse_ms_second_call', float('nan')):.2f} mean4 {k.get('cholesky_inverse_ms_mean_of_4', float('nan')):.2f} | "
      f"eigh {o['c4_ekfac_resnet18']['eigh_ms']:.1f} ekfac {o['c4_ekfac_resnet18']['ekfac_total_ms']:.1f} | kfac matvec {k.get('matvec_ms', float('nan'))}")
PY
  done
done
