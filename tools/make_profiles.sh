# Regenerates profiles/r01_* on the GPU box (run via gpurun from the repo root).
set -x
R=$PWD; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-extras --steps 100 --warmup 10"
rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o ks -- $CMD > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks/ks_results.db $OUT/r01_c2_n8_bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --steps 100 --warmup 10  (C2, 8 rows/GPU, 1 GPU)"
python $R/tools/gap_analysis.py /tmp/p_ks/ks_results.db head_bwd > $OUT/r01_c2_n8_kernel_chain.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py --no-extras --steps 50 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py --no-extras --steps 50 --warmup 5 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db $OUT/r01_c2_n8_pmc_traffic.json $OUT/r01_c2_n8_pmc_traffic.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --no-extras --steps 50 --warmup 5  (C2, 8 rows/GPU)"
cd $R
cp $OUT/r01_c2_n8_pmc_traffic.json profiles/   # so that the bench's traffic leg reads the fresh numbers
python bench.py > $OUT/r01_bench_n1.json 2> $OUT/bench_stderr.txt
tail -c 3000 $OUT/r01_bench_n1.json
# large-batch points of the same workload (kernel path only)
python tools/probe_c2.py 1 8 9 16 32 64 128 256 512 1024 > $OUT/r01_c2_batch_sweep.txt 2>&1
cd /tmp
for N in 128 512; do
  rocprofv3 --kernel-trace --stats -d /tmp/p_n$N -o k -- python $R/tools/probe_c2.py $N > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_n$N/k_results.db $OUT/r01_c2_n${N}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_c2.py $N  (C2 GGN matvec, $N rows, kernel path)"
done
cd $R
python benchmarks/bench_kfac.py resnet18 > $OUT/r01_kfac_resnet18_b512.json 2>/dev/null
python benchmarks/bench_kfac.py lenet > $OUT/r01_kfac_lenet_b1024.json 2>/dev/null
python tools/probe_gemm.py > $OUT/r01_gemm_f32_shapes.txt 2>&1
ls -la $OUT
cd $R
python benchmarks/bench_kfac.py resnet18 --ekfac > $OUT/r01_kfac_resnet18_b512.json 2>/dev/null
python benchmarks/bench_general.py resnet18 > $OUT/r01_general_resnet18_b512.json 2>/dev/null
python benchmarks/bench_general.py encoder > $OUT/r01_general_encoder_c5.json 2>/dev/null
python tools/probe_cols.py 8 32 64 > $OUT/r01_c2_columns.txt 2>&1
python benchmarks/bench_kfac.py encoder > $OUT/r01_kfac_encoder_b8.json 2>/dev/null
# practical read / write stream ceilings of this GPU (size, occupancy, nt, LDS-DMA)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stream2_bench tools/ubench/stream2.hip && /tmp/stream2_bench > $OUT/r01_ubench_read_stream.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stream3_bench tools/ubench/stream3.hip && /tmp/stream3_bench > $OUT/r01_ubench_write_stream.txt
python tools/probe_hessian.py > $OUT/r01_c2_hessian.txt 2>/dev/null
python tools/probe_cg.py > $OUT/r01_c2_cg.txt 2>/dev/null
python benchmarks/bench_kfac.py lenet --fisher type-2 > $OUT/r01_kfac_lenet_b1024_type2.json 2>/dev/null
python tools/probe_syrk_skinny.py > $OUT/r01_gram_tall_shapes.txt 2>/dev/null
python tools/probe_c1.py 2>/dev/null | grep -v amdgpu > $OUT/r01_c1_matvec.txt
python tools/probe_chol.py 2>/dev/null | grep -v amdgpu > $OUT/r01_cholesky_inverse_sizes.txt
python tools/probe_mlp_zoo.py 2>/dev/null | grep -v amdgpu > $OUT/r01_mlp_shapes.txt
python tools/probe_kfoc.py 2>/dev/null | grep -v amdgpu > $OUT/r01_kfoc_build.txt
python tools/probe_eigh_batched.py 2>/dev/null | grep "n=" > $OUT/r01_eigh_batched.txt
python benchmarks/bench_kfac.py encoder --ekfac > $OUT/r01_kfac_encoder_b8.json 2>/dev/null
python tools/probe_chol_batched.py 2>/dev/null | grep "n=" > $OUT/r01_cholesky_inverse_batched.txt
