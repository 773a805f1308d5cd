out=gpurun_out/cols_r4c; mkdir -p $out; rm -f $out/cols.txt
for v in main u1 u2 u8; do
  lib=$PWD/curvlinops_amd/lib/variants/libclo_$v.so; [ $v = main ] && lib=$PWD/curvlinops_amd/lib/libclo_hip.so
  echo "=== $v" >> $out/cols.txt; CLO_HIP_LIB=$lib python tools/probe_cols.py 32 64 2>&1 | grep "K=" >> $out/cols.txt
done
for b in 1 2 3 4 6 8; do
  echo "=== main CLO_KC_BPC=$b" >> $out/cols.txt; CLO_KC_BPC=$b python tools/probe_cols.py 32 64 2>&1 | grep "K=[36]" >> $out/cols.txt
done
cat $out/cols.txt
