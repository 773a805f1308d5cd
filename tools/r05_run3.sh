R=$PWD; OUT=$R/gpurun_out/r05_run3; mkdir -p $OUT
export TMPDIR=/tmp
python tools/stress_kfac_capture.py 4 > $OUT/stress_q4.txt 2>&1; tail -12 $OUT/stress_q4.txt
python tools/stress_kfac_capture.py 16 > $OUT/stress_q16.txt 2>&1; tail -12 $OUT/stress_q16.txt
python -m pytest tests/test_nets.py tests/test_gpu_kernels.py -x -q -m gpu -k "pixel or captured or fused_patch or eigh" > $OUT/new_tests.txt 2>&1; tail -6 $OUT/new_tests.txt
cd /tmp; export MIOPEN_FIND_MODE=FAST
for q in 4 16; do
rm -rf /tmp/pkb$q
GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace -d /tmp/pkb$q -o k -- python $R/tools/prof_kfac_build.py > /dev/null 2>&1
{ echo "# GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace -- python tools/prof_kfac_build.py  (ResNet-18, C4: 512 rows, joint W+b, 1 MC sample; 4 warm-up builds,"
  echo "# MIOPEN_FIND_MODE=FAST; the section between two marker launches = ONE warm build = one replay of the captured graph; tools/kfac_trace_summary.py)"
  python $R/tools/kfac_trace_summary.py /tmp/pkb$q/k_results.db 512; } > $OUT/kfac_build_kernels_q$q.txt
done
cat $OUT/kfac_build_kernels_q4.txt
