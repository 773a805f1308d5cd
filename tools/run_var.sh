# A/B of library variants (curvlinops_amd/lib/variants/libclo_<name>.so, see tools/buildvar.sh) on the C2 8-row chain
echo "--- default"; python tools/probe_c2.py 8 8
for v in "$@"; do echo "--- $v"; CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so python tools/probe_c2.py 8; done

