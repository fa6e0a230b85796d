set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -25
timeout 300 python tools/fft_probe.py > gpurun_out/fft_probe.txt 2>&1; cat gpurun_out/fft_probe.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bank7.json 2> gpurun_out/bank7.err; tail -c 700 gpurun_out/bank7.json
