R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r12
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pgk -o g -- python $R/tools/r6/probe_gemm_kernel_time.py > /tmp/gk.log 2>&1
grep SHAPE /tmp/gk.log > /tmp/gk_shapes.txt
python $R/tools/r6/gemm_kernel_time_summary.py /tmp/pgk/g_results.db /tmp/gk_shapes.txt > $R/gpurun_out/r12/gemm_kernel_time.txt 2>&1
cat $R/gpurun_out/r12/gemm_kernel_time.txt | cut -c1-330
