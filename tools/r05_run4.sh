R=$PWD; OUT=$R/gpurun_out/r05_run4; mkdir -p $OUT
export TMPDIR=/tmp
python tools/diag_kfac_bimodal.py 16 > $OUT/bimodal_q16.txt 2>&1; tail -5 $OUT/bimodal_q16.txt
python tools/diag_kfac_bimodal.py 4 > $OUT/bimodal_q4.txt 2>&1; tail -5 $OUT/bimodal_q4.txt
for q in 4 8 16; do python tools/probe_queues.py $q 2>&1 | grep queues= >> $OUT/queues.txt; done; cat $OUT/queues.txt
python -m pytest tests/test_nets.py tests/test_gpu_kernels.py tests/test_operators_gpu.py -x -q -m gpu -k "pixel or captured or fused_patch or eigh or grouped or equal_shape" > $OUT/new_tests.txt 2>&1; tail -6 $OUT/new_tests.txt
