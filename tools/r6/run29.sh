cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_nets.py -m gpu -q -x 2>&1 | tail -3
python tools/probe_c2.py 64 65 80 96 112 128 129 2>&1 | grep "N="
