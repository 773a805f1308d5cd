# rocprofv3 kernel trace of a few KFAC factor builds (ResNet-18, 512 rows) + per-stream timeline of the last one
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/probe_kfac_build.py > /tmp/kt.log 2>&1
tail -3 /tmp/kt.log
python $GRAFT_REPO_ROOT/tools/kfac_timeline.py /tmp/kt/kt_results.db
