R=$PWD; python tools/probe_cols.py 4 8 32 64
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pc -o k -- python $R/tools/probe_cols.py 32 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/pc/k_results.db /tmp/pc/sum.txt "cols"; head -20 /tmp/pc/sum.txt | cut -c1-150
