out=gpurun_out/r4e; mkdir -p $out
for v in q0p1 q0p3 q1p3 q1p6 q1p3pre0 q1p3pre4; do
  echo "=== $v" >> $out/ab.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so VARIANTS="FLAGS=0" timeout 300 python tools/probe_chain_ab.py >> $out/ab.txt 2>&1
done
for v in t_q1p3; do
  echo "=== $v" >> $out/timeline.txt
  STAMPS_OUT=$out/stamps_$v.npy CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
grep -E "===|round 2|rel diff" $out/ab.txt
cat $out/timeline.txt
