# C2 persistent kernel against its own skeletons (-DCLO_MG_ABLATE bits: 1 MFMAs of layers 1 / 2, 2 delta_1 sweep, 4 out_W2 FMAs,
# 8 out_W1 FMAs; 15 = no arithmetic at all: loads, LDS copies, seams and stores only)
for v in default skel1 skel2 skel12 skel15; do
  if [ $v = default ]; then python tools/probe_mega_variant.py; else CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so python tools/probe_mega_variant.py; fi
done 2>&1 | grep "us per product"
