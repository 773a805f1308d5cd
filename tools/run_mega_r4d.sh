out=gpurun_out/r4d; mkdir -p $out
for v in polls1 polls4 polls8; do
  echo "=== $v" >> $out/ab.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so VARIANTS="FLAGS=0" timeout 300 python tools/probe_chain_ab.py >> $out/ab.txt 2>&1
done
for v in t_polls4; do
  echo "=== $v" >> $out/timeline.txt
  STAMPS_OUT=$out/stamps_$v.npy CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
grep -E "===|round|rel diff" $out/ab.txt
cat $out/timeline.txt
