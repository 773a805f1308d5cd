R=$PWD; OUT=$R/gpurun_out/r05_run8; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fails_soft or eigh" > $OUT/failsoft_eigh.txt 2>&1; tail -15 $OUT/failsoft_eigh.txt
python -m pytest tests -x -q -m gpu > $OUT/suite.txt 2>&1; tail -4 $OUT/suite.txt
