# rocprofv3 kernel trace of ONE warm ResNet-18 KFAC factor build (+ inverses) -> gpurun_out/profiles/r03_kfac_resnet18_*
R=$PWD; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
rm -rf /tmp/pkb
WITH_INVERSE=1 rocprofv3 --kernel-trace -d /tmp/pkb -o k -- python $R/tools/prof_kfac_build.py > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace -- python tools/prof_kfac_build.py  (ResNet-18, C4: 512 rows, joint W+b, 1 MC sample; 4 warm-up builds,"
  echo "# MIOPEN_FIND_MODE=FAST; the section between two marker launches = ONE warm build; tools/kfac_trace_summary.py)"
  python $R/tools/kfac_trace_summary.py /tmp/pkb/k_results.db 512; } > $OUT/r03_kfac_resnet18_build_kernels.txt
cat $OUT/r03_kfac_resnet18_build_kernels.txt
