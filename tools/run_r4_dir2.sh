#!/bin/bash
# round 4: directional with colorless terms at full size, capture-probe node count, bench directional roofline on k_em_bwd
set -x -o pipefail
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -k "directional" -x -q 2>&1 | tail -15 &&
timeout -k 10 600 python -m pytest tests/test_gpu_bank.py -k "allreduce_captured" -x -q -rs 2>&1 | tail -8 &&
timeout -k 10 300 python bench.py --config directional --steps 40 > gpurun_out/r04_dir_new.json 2> gpurun_out/r04_dir_new.err
tail -c 3000 gpurun_out/r04_dir_new.json | head -c 1500; tail -5 gpurun_out/r04_dir_new.err
