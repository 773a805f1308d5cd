out=gpurun_out/r4b; mkdir -p $out
for v in p3 p1 p2pre0 p3pre0 p3pre1; do
  echo "=== $v" >> $out/ab.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so VARIANTS="X=1" timeout 300 python tools/probe_chain_ab.py >> $out/ab.txt 2>&1
done
for v in t_p3 t_p1 t_p3pre0; do
  echo "=== $v" >> $out/timeline.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
grep -E "===|round" $out/ab.txt
cat $out/timeline.txt
