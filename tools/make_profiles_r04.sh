# Round-4 profiles on the GPU box (run via gpurun from the repo root): everything under gpurun_out/profiles_r04/; the files
# that should be judged are then copied into profiles/.  C2 passes use the DRIVER's command (--steps 20 --warmup 5).
R=$PWD; OUT=$R/gpurun_out/profiles_r04; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/make_profiles_c2_r04.sh > $OUT/c2_log.txt 2>&1
# ---- K = 32 probe columns: kernel stats and PMC traffic (FETCH / WRITE in separate passes)
cd /tmp
rm -rf /tmp/pk32 /tmp/pk32f /tmp/pk32w
rocprofv3 --kernel-trace --stats -d /tmp/pk32 -o k -- python $R/tools/probe_cols.py 32 > $OUT/k32_probe.txt 2>&1
python $R/tools/prof_summary.py /tmp/pk32/k_results.db $OUT/r04_c2_k32_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/probe_cols.py 32  (C2, 8 rows, K = 32 columns through the operator API: 2 warm-up + 6 timed products; the mlp_mega rows are the 55 single-vector products of the same script)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pk32f -o f -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pk32w -o w -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pk32f/f_results.db /tmp/pk32w/w_results.db $OUT/r04_c2_k32_pmc_traffic.json $OUT/r04_c2_k32_pmc_traffic.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/probe_cols.py 32  (C2, 8 rows, K = 32: bytes per launch; one product = 3 kfwd_stream + 3 kouter_stream launches + the GEMM chain; algorithmic 8 D K = 2563 MB per product)"
cd $R
grep "K=" $OUT/k32_probe.txt | grep -v amdgpu > $OUT/r04_c2_columns.txt
python tools/probe_cols.py 8 32 64 2>&1 | grep "K=" >> $OUT/r04_c2_columns.txt
# ---- KFAC factor build (ResNet-18) kernel table, EKFAC correction pass
cd /tmp; export MIOPEN_FIND_MODE=FAST
rm -rf /tmp/pkb
WITH_INVERSE=1 rocprofv3 --kernel-trace -d /tmp/pkb -o k -- python $R/tools/prof_kfac_build.py > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python tools/prof_kfac_build.py  (ResNet-18, C4: 512 rows, joint W+b, 1 MC sample; 4 warm-up builds,"
  echo "# MIOPEN_FIND_MODE=FAST; the section between two marker launches = ONE warm build; tools/kfac_trace_summary.py)"
  python $R/tools/kfac_trace_summary.py /tmp/pkb/k_results.db 512; } > $OUT/r04_kfac_resnet18_build_kernels.txt
cd $R
bash tools/run_prof_ekfac3.sh > $OUT/r04_ekfac_correction_kernels.txt 2>&1
unset MIOPEN_FIND_MODE
# ---- eigensolver: reduction / full eigh per size, workgroup caps, EKFAC basis set over worker streams
{ echo "# python tools/probe_sytrd_r4.py (clo_sytrd_f32 with persistent panel launches; PSD factors X^T X / r normalised to max |A| = 1:"
  echo "# lowrank r = n / 3, wishart r = 2 n; eigh = linalg_native.eigh incl. verification -- the wishart rows fail the 1e-3 absolute"
  echo "# residual check of the normalised matrix (|A|_2 = 0.3 n |A|max) and include the float64 retry)"
  for mb in 0 128 64; do echo "--- max_blocks $mb"; MAXB=$mb python tools/probe_sytrd_r4.py 577 1153 2305 4609 2>&1 | grep "n="; done
  echo "# python tools/probe_eigh_streams.py 1 2 4 6 8   (ResNet-18 C4 factors, 512 rows: eigh_many of the 42 factors)"
  python tools/probe_eigh_streams.py 1 2 4 6 8 2>&1 | grep -v amdgpu; } > $OUT/r04_eigh_persistent_sytrd.txt
# ---- batch sweep of the C2 matvec
python tools/probe_c2.py 1 8 9 16 32 33 48 64 65 128 256 512 1024 2>&1 | grep "N=" > $OUT/r04_c2_batch_sweep.txt
# ---- the driver's line (full extras)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_n1.json 2> $OUT/bench_stderr.txt
tail -c 3000 $OUT/r04_bench_n1.json
ls -la $OUT
