cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in default nt2s; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; else unset CLO_HIP_LIB; fi
  echo "== $v"; python tools/probe_c2.py 17 24 32 2>&1 | grep "N="
done
done
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_nt2s.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "mid_rows" 2>&1 | tail -2
