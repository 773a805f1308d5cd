out=gpurun_out/eigh_r4; mkdir -p $out
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q -k "sytrd or eigh or tridiag or reflector or ekfac or persistent" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 900 python tools/probe_eigh_streams.py 1 2 3 4 6 8 > $out/streams.txt 2>&1; grep -v amdgpu $out/streams.txt
