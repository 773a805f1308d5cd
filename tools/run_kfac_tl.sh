R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pktl -o k -- python $R/tools/probe_kfac_build.py > /dev/null 2>&1
python $R/tools/kfac_timeline.py /tmp/pktl/k_results.db
