cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "rows_chain or rows_path" 2>&1 | tail -2
for rep in 1 2; do
for v in default nopad32; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; else unset CLO_HIP_LIB; fi
  echo "== $v"; python tools/probe_c2.py 64 65 72 80 96 97 2>&1 | grep "N="
done
done
