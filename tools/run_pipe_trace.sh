cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p_stats
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/p_stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --pipe-steps ${1:-4} > $GRAFT_REPO_ROOT/gpurun_out/p_stats.log 2>&1)
python tools/timeline_raw.py gpurun_out/p_stats 0.5 150 > gpurun_out/p_timeline.txt
