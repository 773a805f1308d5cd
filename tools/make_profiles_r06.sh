# Round-6 profiles on the GPU box (run via gpurun from the repo root): everything under gpurun_out/profiles_r06/; the files
# that should be judged are then copied into profiles/.  C2 passes use the DRIVER's command (--steps 20 --warmup 5).
R=$PWD; OUT=$R/gpurun_out/profiles_r06; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --gpus 1 --no-extras --steps 20 --warmup 5"
rm -rf /tmp/p_ks /tmp/p_f /tmp/p_w
rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o ks -- $CMD > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks/ks_results.db $OUT/r06_c2_n8_bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --no-extras --steps 20 --warmup 5  (C2, 8 rows/GPU, 1 GPU; the driver's command)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db $OUT/r06_c2_n8_pmc_traffic.json $OUT/r06_c2_n8_pmc_traffic.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --gpus 1 --no-extras --steps 20 --warmup 5  (C2, 8 rows/GPU)"
# ---- K = 32 probe columns
rm -rf /tmp/pk32 /tmp/pk32f /tmp/pk32w
rocprofv3 --kernel-trace --stats -d /tmp/pk32 -o k -- python $R/tools/probe_cols.py 32 > $OUT/k32_probe.txt 2>&1
python $R/tools/prof_summary.py /tmp/pk32/k_results.db $OUT/r06_c2_k32_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_cols.py 32  (C2, 8 rows, K = 32 columns through the operator API: 2 warm-up + 6 timed products; the mlp_mega rows are the 55 single-vector products of the same script)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pk32f -o f -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pk32w -o w -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pk32f/f_results.db /tmp/pk32w/w_results.db $OUT/r06_c2_k32_pmc_traffic.json $OUT/r06_c2_k32_pmc_traffic.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/probe_cols.py 32  (C2, 8 rows, K = 32: bytes per launch; algorithmic 8 D K = 2563 MB per product)"
# ---- KFAC factor build (ResNet-18): one replay of the captured graph, and the eager build; inverses behind it
export MIOPEN_FIND_MODE=FAST
rm -rf /tmp/pkb /tmp/pkbe
WITH_INVERSE=1 rocprofv3 --kernel-trace -d /tmp/pkb -o k -- python $R/tools/prof_kfac_build.py > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python tools/prof_kfac_build.py  (ResNet-18, C4: 512 rows, joint W+b, 1 MC sample; 4 warm-up builds,"
  echo "# MIOPEN_FIND_MODE=FAST; the section between two marker launches = ONE warm build = one replay of the captured hipGraph"
  echo "# (two branches at the default 4 hardware queues: input covariances on the factor stream behind one event; the gradient"
  echo "# covariances of all layers in ONE grouped launch, clo::syrk_grouped_kernel, at the end of the backward pass); tools/kfac_trace_summary.py)"
  python $R/tools/kfac_trace_summary.py /tmp/pkb/k_results.db 512; } > $OUT/r06_kfac_resnet18_build_kernels.txt
KFAC_EAGER=1 rocprofv3 --kernel-trace -d /tmp/pkbe -o k -- python $R/tools/prof_kfac_build.py > /dev/null 2>&1
{ echo "# the same with computers._CAPTURE = False (eager build: one fork of the factor stream per hook, host-dispatch bound)"
  python $R/tools/kfac_trace_summary.py /tmp/pkbe/k_results.db 512; } > $OUT/r06_kfac_resnet18_build_kernels_eager.txt
# ---- C5: who owns hutchpp_trace(96)?
rm -rf /tmp/pc5
rocprofv3 --kernel-trace -d /tmp/pc5 -o k -- python $R/tools/prof_c5_hutchpp.py > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python tools/prof_c5_hutchpp.py  (C5: 12-layer d = 768 encoder, D = 85 M, 8 x 128 tokens;"
  echo "# ONE warm hutchpp_trace(EFLinearOperator, 96 products) between two marker launches: clo:: kernels vs the framework's)"
  SECTION_TITLE="hutchpp_trace(96)" python $R/tools/kfac_trace_summary.py /tmp/pc5/k_results.db 8; } > $OUT/r06_c5_hutchpp_kernel_split.txt
unset MIOPEN_FIND_MODE
cd $R
# ---- eigensolver per size (low-rank and full-rank factors), fp64 retries counted
{ echo "# python tools/probe_sytrd_r4.py (clo_sytrd_f32 persistent panels; PSD factors X^T X / r normalised to max |A| = 1: lowrank r = n / 3,"
  echo "# wishart r = 2 n; eigh = linalg_native.eigh incl. verification -- round 5: the residual test is relative to ||A||_F, full-rank"
  echo "# factors are no longer redone in float64)"
  MAXB=0 python tools/probe_sytrd_r4.py 577 1153 2305 4609 2>&1 | grep "n="
  echo "# python tools/diag_eigh_verify.py: residual of the float32 result in units of eps32 ||A||"
  python tools/diag_eigh_verify.py 577 1153 2305 4609 2>&1 | grep "n="; } > $OUT/r06_eigh_persistent_sytrd.txt
# ---- batch sweep, columns, skeleton
python tools/probe_c2.py 1 8 9 16 17 32 33 48 64 65 128 256 512 1024 2>&1 | grep "N=" > $OUT/r06_c2_batch_sweep.txt
cd /tmp
for n in 16 32 64; do
rm -rf /tmp/pr$n
rocprofv3 --kernel-trace --stats -d /tmp/pr$n -o k -- python $R/tools/probe_c2.py $n > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/pr$n/k_results.db $OUT/r06_c2_n${n}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_c2.py $n  (C2 GGN matvec, $n rows: the 9 ... 64-row MFMA chain, three-product forward)"
done
cd $R
{ echo "# python tools/probe_gemm_sweep_r5.py (round 6 library; clo = the engine's automatic choice, torch = hipBLASLt)"
  python tools/probe_gemm_sweep_r5.py 2>&1 | grep -v amdgpu; } > $OUT/r06_gemm_midsize_sweep.txt
{ echo "# python tools/r6/probe_syrk_grouped.py: the 20 gradient covariances of a ResNet-18 batch (512 rows), one grouped launch vs one clo_syrk_accum_f32 each"
  python tools/r6/probe_syrk_grouped.py 2>&1 | grep -v amdgpu; } > $OUT/r06_syrk_grouped.txt
{ echo "# python tools/probe_kfac_inverse.py: K.inverse(damping) of ResNet-18's 42 factors, six calls (first includes workspace allocation), ms"
  python tools/probe_kfac_inverse.py 2>&1 | grep -v amdgpu
  GPU_MAX_HW_QUEUES=16 python tools/probe_kfac_inverse.py 2>&1 | grep -v amdgpu | sed 's/^/GPU_MAX_HW_QUEUES=16 /'; } > $OUT/r06_cholesky_42_inverses.txt
python tools/probe_cols.py 8 32 64 2>&1 | grep "K=" > $OUT/r06_c2_columns.txt
python tools/probe_fold.py 2>&1 | grep -v amdgpu > $OUT/r06_kfac_factor_kernels_per_shape.txt
for q in 4 16; do python tools/probe_kfac_fork.py $q 2>&1 | grep queues=; done > $OUT/r06_kfac_capture_fork_modes.txt
# ---- the driver's line (full extras), twice; the traffic / kernel summaries of THIS library first, so the line cites them
cp $OUT/r06_c2_n8_pmc_traffic.json $OUT/r06_c2_n8_pmc_traffic.txt $OUT/r06_c2_n8_bench_kernel_stats.txt $R/profiles/
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_n1.json 2> $OUT/bench_stderr.txt
python bench.py --gpus 1 > $OUT/r06_bench_n1_default_steps.json 2>> $OUT/bench_stderr.txt
tail -c 1500 $OUT/r06_bench_n1.json
ls -la $OUT
