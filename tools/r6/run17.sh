cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "mid_rows_chain or ggn_matvec" > $O/t.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -6 $O/t.log
timeout 300 python tools/probe_c2.py 9 16 17 32 33 48 64 2>&1 | grep "N=" | sed 's/^/fused   /' | tee -a $O/sweep.txt
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_nofuse.so timeout 300 python tools/probe_c2.py 9 16 17 32 33 48 64 2>&1 | grep "N=" | sed 's/^/2-launch /' | tee -a $O/sweep.txt
