out=gpurun_out/r4h; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "persistent or mega or ggn_matvec" > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
echo "=== default" >> $out/ab.txt
timeout 300 python tools/probe_chain_ab.py >> $out/ab.txt 2>&1
for v in t_tail; do
  echo "=== $v" >> $out/timeline.txt
  CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_$v.so timeout 300 python tools/probe_mega_timing.py >> $out/timeline.txt 2>&1
done
grep -E "===|round 2|rel diff" $out/ab.txt
cat $out/timeline.txt
