# per-kernel durations / gaps of the C2 8-row chain under rocprofv3 for a list of env settings
# usage: bash tools/run_chain_prof.sh "CLO_MLP_CHAIN4=0" "CLO_MLP_CHAIN4=1" ...   (ONCE=kernel launched once per matvec)
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for V in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pc$i
  env $V rocprofv3 --kernel-trace -d /tmp/pc$i -o k -- python $R/tools/probe_c2.py ${ROWS:-8} > /dev/null 2>&1
  echo "=== $V"; python $R/tools/gap_analysis.py /tmp/pc$i/k_results.db ${ONCE:-outer_all}
done
