# One gpurun call for the directional configuration's records in profiles/:  bash tools/run_dir_measurements.sh [tag]
set -x
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_dir_stats $OUT/${TAG}_dir_pmc_fetch $OUT/${TAG}_dir_pmc_write
timeout 500 python bench.py --config directional > $OUT/${TAG}_directional_bench.json 2> $OUT/${TAG}_directional_bench.err && tail -c 1200 $OUT/${TAG}_directional_bench.json && \
cd /tmp && export TMPDIR=/tmp && \
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_dir_stats -- python $GRAFT_REPO_ROOT/bench.py --config directional --no-cpu-baseline --steps 50 > $OUT/${TAG}_dir_stats.log 2>&1 && \
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_dir_pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --config directional --no-cpu-baseline --steps 2 --warmup 1 --bands 1 > /dev/null 2>&1 && \
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_dir_pmc_write -- python $GRAFT_REPO_ROOT/bench.py --config directional --no-cpu-baseline --steps 2 --warmup 1 --bands 1 > /dev/null 2>&1 && \
ls $OUT/${TAG}_dir_pmc_fetch/*/ | head
