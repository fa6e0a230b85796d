set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bank.py -x -q 2>&1 | tail -15 || exit 1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bank7.json 2> gpurun_out/bank7.err; tail -c 1800 gpurun_out/bank7.json; tail -5 gpurun_out/bank7.err
timeout 300 python bench.py --no-cpu-baseline --bands 1 > gpurun_out/bank1.json 2> gpurun_out/bank1.err; tail -c 600 gpurun_out/bank1.json
