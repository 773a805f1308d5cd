# round 5, first GPU pass: liveness tests, eigh residual scale, sytrd probe, full suite, driver bench line
R=$PWD; OUT=$R/gpurun_out/r05_run1; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_operators_gpu.py -x -q -m gpu -k "updates or qualifying or jacobian_operators_follow or equal_shape or grouped" > $OUT/live_tests.txt 2>&1; tail -5 $OUT/live_tests.txt
python tools/diag_eigh_verify.py 577 1153 2305 4609 2>&1 | grep "n=" > $OUT/eigh_residual_scale.txt; cat $OUT/eigh_residual_scale.txt
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eigh" > $OUT/eigh_tests.txt 2>&1; tail -5 $OUT/eigh_tests.txt
{ for mb in 0; do echo "--- max_blocks $mb"; MAXB=$mb python tools/probe_sytrd_r4.py 577 1153 2305 4609 2>&1 | grep "n="; done; } > $OUT/eigh_sytrd.txt; cat $OUT/eigh_sytrd.txt
python tools/probe_kfac_capture.py 512 > $OUT/kfac_capture.txt 2>&1; tail -5 $OUT/kfac_capture.txt
python -m pytest tests -x -q -m gpu > $OUT/suite.txt 2>&1; tail -5 $OUT/suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_stderr.txt; tail -c 2500 $OUT/bench_n1.json
