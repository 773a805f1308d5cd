# product time of a list of persistent-kernel variants (curvlinops_amd/lib/variants/libclo_<name>.so), two rounds
out=gpurun_out/r5_sseam; mkdir -p $out
for r in 1 2; do for v in $VARIANTS; do
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_variant.py
done; done 2>&1 | grep "us per" | tee -a $out/times.txt
