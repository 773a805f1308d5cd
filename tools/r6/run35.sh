cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_operators_gpu.py -m gpu -q -x -k "chol or inverse or kfac or potrf" 2>&1 | tail -2
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_potrftime.so python tools/r6/probe_potrf_timeline.py 2>&1 | grep -v amdgpu
python tools/probe_kfac_inverse.py 2>&1 | grep -v amdgpu | tail -2
