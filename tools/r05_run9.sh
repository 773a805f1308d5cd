R=$PWD; OUT=$R/gpurun_out/r05_run9; mkdir -p $OUT
export TMPDIR=/tmp
python tools/probe_bench_kfac_leg.py 2>&1 | grep -v amdgpu > $OUT/kfac_leg_alone.txt; cat $OUT/kfac_leg_alone.txt
python tools/probe_bench_kfac_leg.py after 2>&1 | grep -v amdgpu > $OUT/kfac_leg_after.txt; cat $OUT/kfac_leg_after.txt
