set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -25
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bank7.json 2> gpurun_out/bank7.err; tail -c 900 gpurun_out/bank7.json
bash tools/run_trace.sh
