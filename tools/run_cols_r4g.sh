out=gpurun_out/cols_r4g; mkdir -p $out; rm -f $out/cols.txt
R=$PWD
timeout 600 python -X faulthandler -m pytest tests -m gpu -x -q -k "matmat or cols or column" 2>&1 | tail -2
for v in main kdoff kdoff4 kdw8 kdoffu; do
  lib=$R/curvlinops_amd/lib/variants/libclo_$v.so; [ $v = main ] && lib=$R/curvlinops_amd/lib/libclo_hip.so
  echo "=== $v" >> $out/cols.txt; CLO_HIP_LIB=$lib timeout 120 python tools/probe_cols.py 32 64 2>&1 | grep "K=[36]" >> $out/cols.txt
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc_$v && CLO_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pc_$v -o k -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1; python $R/tools/prof_summary.py /tmp/pc_$v/k_results.db $R/$out/k32_$v.txt "variant $v" )
  grep -E "kfwd" $out/k32_$v.txt | cut -c1-100 >> $out/cols.txt
done
cat $out/cols.txt
