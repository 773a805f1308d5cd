cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run12; mkdir -p $O
python tools/probe_gemm_sweep_r5.py 2>&1 | grep "^M=" > $O/gemm_default.txt
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_small2.so python tools/probe_gemm_sweep_r5.py 2>&1 | grep "^M=" > $O/gemm_small2.txt
paste -d'\n' $O/gemm_default.txt $O/gemm_small2.txt | cut -c1-140
