R=$PWD; cd /tmp && export TMPDIR=/tmp
for N in "$@"; do
rocprofv3 --kernel-trace -d /tmp/pk$N -o k -- python $R/tools/probe_c2.py $N > /dev/null 2>&1
echo "=== N=$N"; python $R/tools/gap_analysis.py /tmp/pk$N/k_results.db ${ONCE:-loss_hessian}
done
