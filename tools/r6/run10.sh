cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run10; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -8 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
python bench.py --steps 20 --warmup 5 2>$O/bench_err.log > $O/bench.json; tail -c 3000 $O/bench.json
