R=$PWD; OUT=$R/gpurun_out/r05_run13; mkdir -p $OUT
python -m pytest tests/test_operators_gpu.py tests/test_gpu_kernels.py tests/test_nets.py -x -q -m gpu -k "kron or Kron or equal_shape or grouped or kfac or KFAC or ekfac or gemm or fuzz" > $OUT/kron_tests.txt 2>&1; tail -4 $OUT/kron_tests.txt
python tools/probe_bench_kfac_leg.py 2>&1 | grep ms_per_batch | tee $OUT/kfac_leg.txt
python tools/probe_kron_blocks.py 2>&1 | grep "@ \[D, 1\]" | tee $OUT/kron_blocks.txt
