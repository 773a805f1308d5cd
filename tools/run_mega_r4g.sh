out=gpurun_out/r4g; mkdir -p $out
for rep in 1 2 3; do
for v in cnt sent; do
  echo "=== $v" >> $out/ab.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so VARIANTS="FLAGS=0" timeout 300 python tools/probe_chain_ab.py >> $out/ab.txt 2>&1
done
done
for v in t_cnt t_sent t_cnt t_sent; do
  echo "=== $v" >> $out/timeline.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
grep -E "===|round 2" $out/ab.txt
grep -E "===|entry|gathered|L2 mfma|end" $out/timeline.txt
