cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r14
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or syrk or kron or chol or eigh or kfac" 2>&1 | tail -5
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_v3time.so timeout 300 python tools/r6/probe_gemm_timeline.py > gpurun_out/r14/gemm_timeline.txt 2>&1
grep -A12 "M=512 N=2304 K=2304\|M=2048\|M=384" gpurun_out/r14/gemm_timeline.txt
for v in wide narrow; do
  if [ $v = narrow ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_v3narrow.so; fi
  echo "== $v"; timeout 600 python tools/probe_gemm_sweep_r5.py 2>&1 | grep -v amdgpu > gpurun_out/r14/sweep_$v.txt; cat gpurun_out/r14/sweep_$v.txt
done
