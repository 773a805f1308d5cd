cd /tmp && export TMPDIR=/tmp
export CLO_HIP_LIB=$GRAFT_REPO_ROOT/curvlinops_amd/lib/variants/libclo_v3time.so
rocprofv3 --kernel-trace --stats -d /tmp/pdbg -o d -- python $GRAFT_REPO_ROOT/tools/r6/dbg_v3time.py > /tmp/dbg.log 2>&1
grep -v "simple_timer\|amdgpu.ids" /tmp/dbg.log | tail -8
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pdbg/d_results.db /tmp/sum.txt "dbg"; cut -c1-180 /tmp/sum.txt | head -14
