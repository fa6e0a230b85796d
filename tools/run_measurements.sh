set -x
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r01_stats gpurun_out/r01_pmc_fetch gpurun_out/r01_pmc_write
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/r01_bench_n1.json 2> gpurun_out/r01_bench_n1.err; tail -c 1500 gpurun_out/r01_bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_fetch/*/ | head
