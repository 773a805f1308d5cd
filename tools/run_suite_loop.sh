# round-4 crash hunt: the full GPU suite N times back to back with faulthandler, every log kept
n=${1:-5}; tag=${2:-loop}; out=gpurun_out/suite_$tag; mkdir -p $out
ulimit -c unlimited
for i in $(seq 1 $n); do
  timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $out/run_$i.log 2>&1
  echo "run $i exit code $?" | tee -a $out/summary.txt
  tail -1 $out/run_$i.log >> $out/summary.txt
  grep -l "Fatal Python error\|Segmentation\|core dumped\|Aborted" $out/run_$i.log >> $out/summary.txt
done
ls core* 2>/dev/null | head >> $out/summary.txt
cat $out/summary.txt
