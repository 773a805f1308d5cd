cd /tmp; export TMPDIR=/tmp
for n in 16 32 48 64; do
rm -rf /tmp/pr$n
rocprofv3 --kernel-trace --stats -d /tmp/pr$n -o k -- python /root/repo/tools/probe_c2.py $n > /dev/null 2>&1
python /root/repo/tools/prof_summary.py /tmp/pr$n/k_results.db /tmp/st$n.txt "n=$n"
head -9 /tmp/st$n.txt | cut -c1-110
done
