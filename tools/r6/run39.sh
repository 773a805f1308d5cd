cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ggn or mlp or rows" 2>&1 | tail -2
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_midtime.so python tools/r6/probe_mid_dprev_timeline.py 2>&1 | grep -v amdgpu
python tools/probe_c2.py 9 16 17 32 33 48 49 64 2>&1 | grep "N="
