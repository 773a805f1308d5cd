# C2 (8 rows, K = 1) profiles of round 4 with the DRIVER's command; outputs under gpurun_out/profiles_r04/
set -x
R=$PWD; OUT=$R/gpurun_out/profiles_r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --gpus 1 --no-extras --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o ks -- $CMD > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks/ks_results.db $OUT/r04_c2_n8_bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --no-extras --steps 20 --warmup 5  (C2, 8 rows/GPU, 1 GPU; the driver's command)"
python $R/tools/gap_analysis.py /tmp/p_ks/ks_results.db mlp_mega > $OUT/r04_c2_n8_kernel_chain.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db $OUT/r04_c2_n8_pmc_traffic.json $OUT/r04_c2_n8_pmc_traffic.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --gpus 1 --no-extras --steps 20 --warmup 5  (C2, 8 rows/GPU)"
cd $R
for i in 1 2 3; do python bench.py --gpus 1 --no-extras --steps 20 --warmup 5 >> $OUT/bench_noextras_x3.txt 2>/dev/null; done
cat $OUT/r04_c2_n8_bench_kernel_stats.txt | head -20
cat $OUT/r04_c2_n8_pmc_traffic.txt
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_noextras_x3.txt
