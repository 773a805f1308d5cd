R=$PWD; cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
CLO_KFAC_FUSED_IM2COL=$mode rocprofv3 --kernel-trace --stats -d /tmp/pkf$mode -o k -- python $R/benchmarks/bench_kfac.py resnet18 > /dev/null 2>&1
echo "=== fused=$mode"; python $R/tools/prof_summary.py /tmp/pkf$mode/k_results.db /tmp/pkf$mode.txt "kfac fused=$mode" ; grep -i "clo::\|im2col" /tmp/pkf$mode.txt | head -14
done
