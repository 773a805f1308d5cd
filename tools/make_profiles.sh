# Regenerates the core profiles/r03_* files on the GPU box (run via gpurun from the repo root):
#   bash tools/make_profiles.sh      -> gpurun_out/profiles/*, copy what should be judged into profiles/
# The C2 kernel stats / PMC passes use the DRIVER's command (--steps 20 --warmup 5).
set -x
R=$PWD; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --gpus 1 --no-extras --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o ks -- $CMD > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks/ks_results.db $OUT/r03_c2_n8_bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --no-extras --steps 20 --warmup 5  (C2, 8 rows/GPU, 1 GPU; the driver's command)"
python $R/tools/gap_analysis.py /tmp/p_ks/ks_results.db mlp_mega > $OUT/r03_c2_n8_kernel_chain.txt
FLAGS_NOTE=chain rocprofv3 --kernel-trace --stats -d /tmp/p_ks0 -o ks -- $CMD > /dev/null 2>&1
python $R/tools/gap_analysis.py /tmp/p_ks0/ks_results.db outer_all > $OUT/r03_c2_n8_kernel_chain_six_launches.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db $OUT/r03_c2_n8_pmc_traffic.json $OUT/r03_c2_n8_pmc_traffic.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --gpus 1 --no-extras --steps 20 --warmup 5  (C2, 8 rows/GPU)"
cd $R
bash tools/run_prof_kfac_build.sh > /dev/null 2>&1
cp $OUT/r03_c2_n8_pmc_traffic.json $OUT/r03_c2_n8_bench_kernel_stats.txt $OUT/r03_kfac_resnet18_build_kernels.txt profiles/   # the bench's traffic / rocprof legs read these
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_n1.json 2> $OUT/bench_stderr.txt
tail -c 1500 $OUT/r03_bench_n1.json
python tools/probe_c2.py 1 8 9 16 32 33 48 64 65 128 256 512 1024 2>&1 | grep "N=" > $OUT/r03_c2_batch_sweep.txt
ls -la $OUT
