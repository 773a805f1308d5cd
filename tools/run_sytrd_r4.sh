out=gpurun_out/sytrd_r4; mkdir -p $out
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q -k "sytrd or eigh or tridiag or reflector" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for mb in 256 128 85 64; do echo "--- MAXB=$mb" >> $out/probe.txt; MAXB=$mb timeout 600 python tools/probe_sytrd_r4.py 1153 2305 4609 2>&1 | grep "n=" >> $out/probe.txt; done; cat $out/probe.txt
for mb in 256 128; do echo "--- phases MAXB=$mb"; CLO_HIP_LIB=$PWD/curvlinops_amd/lib/variants/libclo_tdt.so python tools/probe_sytrd_phases_r4.py 4609 $mb 2>&1 | grep -v amdgpu | head -10; done
