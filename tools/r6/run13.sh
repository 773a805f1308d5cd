cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r13
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or syrk or kron or chol" 2>&1 | tail -5
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_v3time.so timeout 300 python tools/r6/probe_gemm_timeline.py > gpurun_out/r13/gemm_timeline.txt 2>&1
grep -A12 "M=512 N=2304 K=2304\|M=512 N=2304 K=512\|M=512 N=4608 K=4608" gpurun_out/r13/gemm_timeline.txt
timeout 600 python tools/probe_gemm_sweep_r5.py 2>&1 | grep -v amdgpu | head -19 > gpurun_out/r13/sweep.txt; cat gpurun_out/r13/sweep.txt
python bench.py --secondary-only 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in j.get('secondary',j).items() if 'rows' in k})" 
