out=gpurun_out/cols_r4d; mkdir -p $out; rm -f $out/cols.txt
for v in main w4e5 w4e6 w8e5; do
  lib=$PWD/curvlinops_amd/lib/variants/libclo_$v.so; [ $v = main ] && lib=$PWD/curvlinops_amd/lib/libclo_hip.so
  for b in 4 8; do
  echo "=== $v CLO_KC_BPC=$b" >> $out/cols.txt; CLO_KC_BPC=$b CLO_HIP_LIB=$lib python tools/probe_cols.py 32 64 2>&1 | grep "K=[36]" >> $out/cols.txt
  done
done
cat $out/cols.txt
