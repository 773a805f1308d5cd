cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run9; mkdir -p $O
python tools/r6/probe_syrk_grouped.py 2>&1 | grep -v amdgpu | tee -a $O/sg.txt
for v in sgv1 sgpipe; do
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_$v.so python tools/r6/probe_syrk_grouped.py 2>&1 | grep -v amdgpu | tee -a $O/sg.txt
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "syrk_grouped" 2>&1 | tail -3
for w in 3 4 5; do
CLO_INV_WORKERS=$w python tools/probe_kfac_inverse.py 2>&1 | grep -v amdgpu | sed "s/^/workers=$w /" | tee -a $O/inv.txt
done
