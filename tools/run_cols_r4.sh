out=gpurun_out/cols_r4; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "matmat or cols or column" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
echo "=== overlap" > $out/cols.txt; python tools/probe_cols.py 8 32 64 >> $out/cols.txt 2>&1
echo "=== sequential" >> $out/cols.txt; CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_kc0.so python tools/probe_cols.py 8 32 64 >> $out/cols.txt 2>&1
cat $out/cols.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pc -o k -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/pc/k_results.db $R/$out/k32_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_cols.py 32  (C2, 8 rows, K = 32 columns; 2 warm-up + 6 timed products + 55 single-vector products)"
head -24 $R/$out/k32_kernel_stats.txt | cut -c1-160
