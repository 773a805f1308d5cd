R=$PWD; OUT=$R/gpurun_out/r05_run5; mkdir -p $OUT
export TMPDIR=/tmp
for q in 4 16; do python tools/probe_kfac_fork.py $q 2>&1 | grep queues= >> $OUT/fork.txt; done; cat $OUT/fork.txt
python tools/probe_fold.py 2>&1 | grep -v amdgpu > $OUT/fold.txt; cat $OUT/fold.txt
