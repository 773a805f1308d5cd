cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r16
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_v3time.so timeout 300 python tools/r6/probe_gemm_timeline.py > gpurun_out/r16/gemm_timeline.txt 2>&1
cat gpurun_out/r16/gemm_timeline.txt
timeout 600 python tools/probe_gemm_sweep_r5.py 2>&1 | grep -v amdgpu | head -19
