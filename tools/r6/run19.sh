cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r19
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4
for v in default v2narrow; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; fi
  echo "== $v"; python tools/probe_c2.py 65 128 256 512 1024 2>&1 | grep "N="
  timeout 600 python tools/probe_gemm_sweep_r5.py 2>&1 | grep -v amdgpu > gpurun_out/r19/sweep_$v.txt; grep "K=  128\|1024 N= 1024\|stem\|layer1\|layer2 \|layer3" gpurun_out/r19/sweep_$v.txt
  python tools/probe_kfac_inverse.py 2>&1 | grep -v amdgpu | tail -3
done
