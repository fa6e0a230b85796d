set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -5
timeout 300 python tools/kernel_bench.py 2>&1 | head -9 > gpurun_out/kbench.txt; cat gpurun_out/kbench.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bank7.json 2> gpurun_out/bank7.err; tail -c 700 gpurun_out/bank7.json
