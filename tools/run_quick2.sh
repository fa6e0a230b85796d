set -x
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/q_stats
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err && tail -c 600 gpurun_out/q_bench.json && \
cd /tmp && export TMPDIR=/tmp && \
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q_stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/q_stats.log 2>&1 && \
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/gpurun_out/q_stats 60 > $GRAFT_REPO_ROOT/gpurun_out/q_timeline.txt && cat $GRAFT_REPO_ROOT/gpurun_out/q_timeline.txt
