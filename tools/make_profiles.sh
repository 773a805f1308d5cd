# Regenerates the core profiles/r02_* files on the GPU box (run via gpurun from the repo root):
#   bash tools/make_profiles.sh      -> gpurun_out/profiles/*, copy what should be judged into profiles/
set -x
R=$PWD; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-extras --steps 100 --warmup 10"
rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o ks -- $CMD > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks/ks_results.db $OUT/r02_c2_n8_bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --steps 100 --warmup 10  (C2, 8 rows/GPU, 1 GPU)"
python $R/tools/gap_analysis.py /tmp/p_ks/ks_results.db head_bwd > $OUT/r02_c2_n8_kernel_chain.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py --no-extras --steps 50 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py --no-extras --steps 50 --warmup 5 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db $OUT/r02_c2_n8_pmc_traffic.json $OUT/r02_c2_n8_pmc_traffic.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --no-extras --steps 50 --warmup 5  (C2, 8 rows/GPU)"
cd $R
cp $OUT/r02_c2_n8_pmc_traffic.json $OUT/r02_c2_n8_bench_kernel_stats.txt profiles/   # the bench's traffic / rocprof legs read these
python bench.py > $OUT/r02_bench_n1.json 2> $OUT/bench_stderr.txt
tail -c 1500 $OUT/r02_bench_n1.json
python tools/probe_c2.py 1 8 9 16 32 33 48 64 65 128 256 512 1024 2>&1 | grep "N=" > $OUT/r02_c2_batch_sweep.txt
python benchmarks/bench_kfac.py resnet18 --ekfac 2>/dev/null > $OUT/r02_kfac_resnet18_b512.json
python benchmarks/bench_kfac.py lenet 2>/dev/null > $OUT/r02_kfac_lenet_b1024.json
python benchmarks/bench_kfac.py lenet --fisher type-2 2>/dev/null > $OUT/r02_kfac_lenet_b1024_type2.json
./tools/ubench/stream5_bench > $OUT/r02_ubench_load_pattern.txt 2>&1
ls -la $OUT
