# full GPU suite with faulthandler and the complete log kept (round-4 crash hunt: every run of the suite is archived)
tag=${1:-a}; out=gpurun_out/suite_$tag; mkdir -p $out
ulimit -c unlimited
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/pytest_full.log 2>&1
echo "exit code $?" >> $out/pytest_full.log
tail -5 $out/pytest_full.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench_noextras.json 2> $out/bench_stderr.txt
cat $out/bench_noextras.json | head -c 1200
