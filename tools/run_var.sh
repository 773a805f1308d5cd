# A/B of library variants on the C2 8-row chain (old 6-launch chain)
export CLO_MLP_CHAIN1=1
echo "--- default"; python tools/probe_c2.py 8
for v in "$@"; do echo "--- $v"; CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so python tools/probe_c2.py 8; done
