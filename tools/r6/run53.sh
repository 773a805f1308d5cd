cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in default fin2 fin1; do
  if [ $v != default ]; then export CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_$v.so; else unset CLO_HIP_LIB; fi
  echo "== $v"; python tools/probe_c2.py 9 16 17 24 32 2>&1 | grep "N="
done
done
