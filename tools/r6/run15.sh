cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r15
for v in v3small2 v3small2n8 v3n8; do
  export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so
  echo "== $v"; timeout 600 python tools/probe_gemm_sweep_r5.py 2>&1 | grep -v amdgpu > gpurun_out/r15/sweep_$v.txt; cat gpurun_out/r15/sweep_$v.txt
done
