cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run11; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_operators_gpu.py -m gpu -q -x > $O/t.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/t.log
python tools/probe_c2.py 9 16 17 32 33 48 64 2>&1 | grep "N=" | sed 's/^/merged  /' | tee -a $O/sweep.txt
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_nomerge.so python tools/probe_c2.py 9 16 17 32 33 48 64 2>&1 | grep "N=" | sed 's/^/separate /' | tee -a $O/sweep.txt
cd /tmp; rm -rf /tmp/pc5
rocprofv3 --kernel-trace -d /tmp/pc5 -o k -- python $R/tools/prof_c5_hutchpp.py > /dev/null 2>&1
SECTION_TITLE="hutchpp_trace(96)" python $R/tools/kfac_trace_summary.py /tmp/pc5/k_results.db 8 > $O/c5_split.txt; grep "clo::\|==" $O/c5_split.txt | cut -c1-150
