cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ggn or mlp or mid or rows" 2>&1 | tail -3
python tools/probe_c2.py 9 16 17 32 33 48 49 64 2>&1 | grep "N="
