#!/bin/bash
# round 4: the directional step with the output stage in the time domain (csrc/dirlin.hip)
set -x -o pipefail
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -k "directional_output_stage" -x -q 2>&1 | tail -15 &&
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -k "directional" -x -q 2>&1 | tail -15 &&
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "directional or f6" -x -q 2>&1 | tail -8 &&
GFDN_DIR_LINES=0 timeout -k 10 300 python bench.py --config directional --no-cpu-baseline --steps 40 > gpurun_out/r04_dir_old.json 2> gpurun_out/r04_dir_old.err &&
timeout -k 10 300 python bench.py --config directional --no-cpu-baseline --steps 40 > gpurun_out/r04_dir_new.json 2> gpurun_out/r04_dir_new.err
tail -c 600 gpurun_out/r04_dir_old.json; tail -c 600 gpurun_out/r04_dir_new.json; tail -5 gpurun_out/r04_dir_new.err
