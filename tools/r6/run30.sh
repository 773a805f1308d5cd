cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ggn or mlp or rows" 2>&1 | tail -3
for v in default nopad64; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; fi
  echo "== $v"; python tools/probe_c2.py 128 129 160 192 193 256 257 320 2>&1 | grep "N="
done
