cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r23
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_nets.py -m gpu -q -x 2>&1 | tail -4
for v in default outer1; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; fi
  echo "== $v"; python tools/probe_c2.py 9 16 17 32 33 48 49 64 2>&1 | grep "N="
done
