out=gpurun_out/cols_r4b; mkdir -p $out
python - <<'PY' > $out/torch_ceilings.txt 2>&1
import torch, time
big = torch.empty(1 << 28, device="cuda"); big2 = torch.empty(1 << 28, device="cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, fn, bytes_ in [("fill 1 GiB", lambda: big.fill_(1.5), 2**30), ("sum 1 GiB", lambda: big.sum(), 2**30),
                         ("copy 1 GiB -> 1 GiB", lambda: big2.copy_(big), 2**31), ("mul_ in place 1 GiB", lambda: big.mul_(1.0001), 2**31)]:
    s = t(fn); print(f"{name:24s} {s*1e6:8.1f} us  {bytes_/s/1e12:5.2f} TB/s")
PY
cat $out/torch_ceilings.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "matmat or cols or column" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for v in main il0 t0 t2 t4; do
  lib=$PWD/curvlinops_amd/lib/variants/libclo_$v.so; [ $v = main ] && lib=$PWD/curvlinops_amd/lib/libclo_hip.so
  echo "=== $v" >> $out/cols.txt; CLO_HIP_LIB=$lib python tools/probe_cols.py 8 32 64 >> $out/cols.txt 2>&1
done
cat $out/cols.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pc -o k -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/pc/k_results.db $R/$out/k32_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_cols.py 32  (C2, 8 rows, K = 32 columns; 2 warm-up + 6 timed products + 55 single-vector products)"
head -12 $R/$out/k32_kernel_stats.txt | cut -c1-160
