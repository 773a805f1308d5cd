R=$PWD; OUT=$R/gpurun_out/r05_run2; mkdir -p $OUT
export TMPDIR=/tmp
python tools/probe_c2_host.py > $OUT/c2_host.txt 2>&1; cat $OUT/c2_host.txt
python -m pytest tests/test_nets.py tests/test_operators_gpu.py -x -q -m gpu -k "kfac or KFAC or resnet or lenet or conv or cnn" > $OUT/kfac_tests.txt 2>&1; tail -4 $OUT/kfac_tests.txt
python tools/probe_kfac_leg.py 16 > $OUT/kfac_leg_q16.txt 2>&1; cat $OUT/kfac_leg_q16.txt
python tools/probe_kfac_leg.py 4 > $OUT/kfac_leg_q4.txt 2>&1; cat $OUT/kfac_leg_q4.txt
python tools/probe_kron_blocks.py > $OUT/kron_blocks.txt 2>&1; cat $OUT/kron_blocks.txt
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eigh" > $OUT/eigh_tests.txt 2>&1; tail -4 $OUT/eigh_tests.txt
