O=gpurun_out; mkdir -p $O
for D in 2 3 4; do timeout 300 python tools/pipeline_probe.py $D 100 >> $O/r05_g_pipeline_probe.txt 2>&1; done; cat $O/r05_g_pipeline_probe.txt | grep depth
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $GRAFT_REPO_ROOT/$O/r05_g_trace -- python $GRAFT_REPO_ROOT/tools/pipeline_probe.py 3 40 > $GRAFT_REPO_ROOT/$O/r05_g_trace.log 2>&1)
python tools/pipeline_timeline.py $O/r05_g_trace > $O/r05_g_pipeline_timeline.txt 2>&1; head -50 $O/r05_g_pipeline_timeline.txt; rm -rf $O/r05_g_trace
