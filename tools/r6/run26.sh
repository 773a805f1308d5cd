cd $GRAFT_REPO_ROOT
for v in default nofull; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; fi
  echo "== $v"; python tools/probe_c2.py 16 32 33 48 64 2>&1 | grep "N="
done
