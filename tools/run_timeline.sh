# kernel timeline of one replayed step:  bash tools/run_timeline.sh <tag> [bench.py arguments]   -> gpurun_out/<tag>_timeline.txt
set -x
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_trace
cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 200 --repeats 1 "$@" > $OUT/${TAG}_trace.log 2>&1 && \
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/${TAG}_trace 100 > $OUT/${TAG}_timeline.txt && cat $OUT/${TAG}_timeline.txt
rm -rf $OUT/${TAG}_trace
