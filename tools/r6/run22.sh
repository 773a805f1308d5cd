cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r22
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_midtime.so timeout 300 python tools/r6/probe_mid_outer_timeline.py > gpurun_out/r22/mid_outer_timeline.txt 2>&1
cat gpurun_out/r22/mid_outer_timeline.txt
