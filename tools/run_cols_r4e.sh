out=gpurun_out/cols_r4e; mkdir -p $out; rm -f $out/cols.txt
R=$PWD
for v in main d2 d4 tpw2 gsm; do
  lib=$R/curvlinops_amd/lib/variants/libclo_$v.so; [ $v = main ] && lib=$R/curvlinops_amd/lib/libclo_hip.so
  echo "=== $v" >> $out/cols.txt; CLO_HIP_LIB=$lib python tools/probe_cols.py 32 64 2>&1 | grep "K=[36]" >> $out/cols.txt
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc_$v && CLO_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/pc_$v -o k -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1; python $R/tools/prof_summary.py /tmp/pc_$v/k_results.db $R/$out/k32_$v.txt "variant $v" )
  grep -E "kfwd|kouter|gemm_v|splitk" $out/k32_$v.txt | cut -c1-110 >> $out/cols.txt
done
echo "=== gemm shapes, variant gsm" >> $out/cols.txt
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_gsm.so python tools/probe_gemm_kcols.py 2>&1 | grep "x \[" >> $out/cols.txt
cat $out/cols.txt
