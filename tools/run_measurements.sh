# One gpurun call that produces everything profiles/ holds for a round:  bash tools/run_measurements.sh [tag] [notests]
# (steps are chained with &&: nothing runs on the GPU after a failed or timed-out step).  Order: the profiled runs first,
# then the tables bench.py reads (tools/make_profiles.py --pre, on the box), then the bench run itself -- its line is
# computed from this call's own tables.
set -x
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
if [ "$2" != "notests" ]; then timeout -k 10 1100 python -u -m pytest tests -q -m gpu -x > $OUT/${TAG}_suite.log 2>&1 || exit 1; fi
cd /tmp && export TMPDIR=/tmp && \
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_stats.log 2>&1 && \
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --eager > /dev/null 2>&1 && \
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --eager > /dev/null 2>&1 && \
cd $GRAFT_REPO_ROOT && python tools/make_profiles.py $TAG --pre && \
timeout 500 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err && tail -c 1500 $OUT/${TAG}_bench_n1.json
