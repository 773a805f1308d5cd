# scalar-path seam A variants of the C2 persistent kernel (-DCLO_MG_SSEAM=0 paced vector seam, 1 scalar poll + gather,
# 2 = 1 with the publish stores ordered before the bulk loads): product time, then the phase timelines
out=gpurun_out/r5_sseam; mkdir -p $out
for r in 1 2; do for v in ${VARIANTS:-ss0 ss1 ss2}; do
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_variant.py
done; done 2>&1 | grep "us per" | tee $out/times.txt
for v in ${VARIANTS:-ss0 ss1 ss2}; do
  echo "=== ${v}t" >> $out/timeline.txt
  STAMPS_OUT=$out/stamps_$v.npy CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_${v}t.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
cat $out/timeline.txt
