python -m pytest tests/test_gpu_kernels.py tests/test_operators_gpu.py -m gpu -x -q 2>&1 | tail -2
for f in 16 11 8 6; do for v in 0 1 6; do echo "F=$f variant $v"; CLO_MF1_F=$f CLO_MF1=$v python bench.py --no-extras 2>&1 | tail -1 | cut -c60-100; done; done
python tools/probe_layer.py
