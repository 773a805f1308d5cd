cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r11
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_v3time.so timeout 300 python tools/r6/probe_gemm_timeline.py > gpurun_out/r11/gemm_timeline.txt 2>&1
tail -80 gpurun_out/r11/gemm_timeline.txt
