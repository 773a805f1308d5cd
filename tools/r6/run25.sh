cd $GRAFT_REPO_ROOT
for v in default mo2_1 mo2_3 mo2_4 mo2_6 outer1; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; fi
  echo "== $v"; python tools/probe_c2.py 33 48 49 64 2>&1 | grep "N="
done
