cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r18
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -5
for v in default v3narrow v3smallepi; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; fi
  echo "== $v"; python tools/probe_c2.py 65 128 256 512 1024 2>&1 | grep "N="
done
