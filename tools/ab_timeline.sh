# timelines of the replayed step under an environment switch: bash tools/ab_timeline.sh VAR
cd $GRAFT_REPO_ROOT
VAR=$1
for v in 0 1; do
rm -rf gpurun_out/q_stats$v
(cd /tmp && export TMPDIR=/tmp && env $VAR=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q_stats$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/q_stats$v.log 2>&1) && \
python tools/timeline.py gpurun_out/q_stats$v 60 > gpurun_out/q_timeline$v.txt && tail -3 gpurun_out/q_timeline$v.txt
done
