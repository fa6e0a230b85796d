set -x
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bank7.json 2> gpurun_out/bank7.err; tail -c 300 gpurun_out/bank7.json
bash tools/run_trace.sh
