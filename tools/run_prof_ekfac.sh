R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pek -o k -- python $R/tools/prof_ekfac.py > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/pek/k_results.db /tmp/pek.txt "ekfac correction"; head -16 /tmp/pek.txt | cut -c1-150
