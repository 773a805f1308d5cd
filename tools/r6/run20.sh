cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r20
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ggn or mlp or mid or rows" 2>&1 | tail -4
python tools/probe_c2.py 8 9 16 17 32 33 40 48 49 64 65 128 2>&1 | grep "N="
