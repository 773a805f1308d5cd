out=gpurun_out/cols_r4f; mkdir -p $out; rm -f $out/cols.txt
R=$PWD
for v in main w4b2 w4b3 w8b2 w2b4; do
  lib=$R/curvlinops_amd/lib/variants/libclo_$v.so; [ $v = main ] && lib=$R/curvlinops_amd/lib/libclo_hip.so
  echo "=== $v" >> $out/cols.txt
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc_$v && CLO_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/pc_$v -o k -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1; python $R/tools/prof_summary.py /tmp/pc_$v/k_results.db $R/$out/k32_$v.txt "variant $v" )
  grep -E "kfwd" $out/k32_$v.txt | cut -c1-100 >> $out/cols.txt
done
cat $out/cols.txt
