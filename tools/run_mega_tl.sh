out=gpurun_out/r4i; mkdir -p $out
for v in t_tail; do
  echo "=== $v" >> $out/timeline.txt
  STAMPS_OUT=$out/stamps_$v.npy CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
cat $out/timeline.txt
