# round 6, GPU call 4: full GPU suite; C2 row sweep (three-product forward) + WV=4 variant; GEMM mid-size sweep with / without the
# k-group form; grouped gradient covariances in the KFAC build; one bench line
cd /root/repo; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r6_run4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; echo "gpu suite rc=$?" >> $O/summary.txt
tail -15 $O/t_all.log
python tools/probe_c2.py 8 9 16 17 32 33 48 64 65 > $O/c2_sweep_new.txt 2>&1
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_wv4.so python tools/probe_c2.py 9 16 32 33 48 64 > $O/c2_sweep_wv4.txt 2>&1
cat $O/c2_sweep_new.txt $O/c2_sweep_wv4.txt
python tools/probe_gemm_sweep_r5.py > $O/gemm_kg2.txt 2>&1
CLO_HIP_LIB=$R/curvlinops_amd/lib/variants/libclo_kg1.so python tools/probe_gemm_sweep_r5.py > $O/gemm_kg1.txt 2>&1
paste -d'\n' $O/gemm_kg2.txt $O/gemm_kg1.txt | head -80
for g in 1 0; do
  CLO_KFAC_GROUP_G=$g python bench.py --steps 20 --warmup 5 2>>$O/bench_err.log > $O/bench_group$g.json
  python -c "
import json
d=json.loads([l for l in open('$O/bench_group$g.json') if l.startswith('{')][-1]); k=d['kfac']
print('group_g=$g', 'ms_per_step %.4f' % d['ms_per_step'], 'kfac %.2f' % k['ms_per_batch'], 'inv first %.1f second %.1f' % (k['cholesky_inverse_ms_first_call'], k['cholesky_inverse_ms_second_call']), {kk: vv for kk, vv in d.get('other_points', {}).items() if 'c3' in kk or 'c5' in kk or 'hutch' in kk or 'rows' in kk})" | tee -a $O/summary.txt
done
