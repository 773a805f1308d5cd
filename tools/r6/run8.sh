cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run8; mkdir -p $O
python tools/r6/probe_syrk_grouped.py 2>&1 | grep -v amdgpu | tee -a $O/sg.txt
for v in sgv1 sgc32 sgc128 sgc256 sgi8 sgi8c128 sgi2; do
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_$v.so python tools/r6/probe_syrk_grouped.py 2>&1 | grep -v amdgpu | tee -a $O/sg.txt
done
