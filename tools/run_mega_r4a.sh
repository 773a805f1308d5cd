# round-4 A/B of the persistent C2 kernel: paced tile requests (CLO_MG_PACE) and XCD-aware tile mapping (CLO_MG_XCD)
out=gpurun_out/r4a; mkdir -p $out
./tools/ubench/seam_probe > $out/seam_probe.txt 2>&1
for v in base p0x1 p2 p3 p4 p6 p3x0; do
  echo "=== $v" >> $out/ab.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_chain_ab.py >> $out/ab.txt 2>&1
done
for v in t_base t_p3 t_p6; do
  echo "=== $v" >> $out/timeline.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
timeout 900 python -m pytest tests -m gpu -x -q -k "persistent or mega or ggn_matvec" > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
cat $out/seam_probe.txt
grep -E "===|round 2|rel diff" $out/ab.txt
