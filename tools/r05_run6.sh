R=$PWD; OUT=$R/gpurun_out/r05_run6; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/run_mega_skeleton.sh > $OUT/mega_skeleton.txt 2>&1; cat $OUT/mega_skeleton.txt
for q in 4 16; do python tools/probe_kfac_fork.py $q 2>&1 | grep queues= >> $OUT/fork.txt; done; cat $OUT/fork.txt
python tools/probe_fold.py 2>&1 | grep "conv3x3" > $OUT/fold.txt; cat $OUT/fold.txt
python -m pytest tests/test_nets.py tests/test_gpu_kernels.py -x -q -m gpu -k "pixel or captured or fused_patch or kfac" > $OUT/new_tests.txt 2>&1; tail -4 $OUT/new_tests.txt
