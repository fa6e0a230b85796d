#!/bin/bash
# round 4: quick check after a kernel change: EDC / bank tests, three bench runs, one timeline
set -x -o pipefail
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -k "edc" -x -q 2>&1 | tail -3 &&
timeout -k 10 900 python -m pytest tests/test_gpu_bank.py -x -q -k "oracle or stored or spectral or distinct" 2>&1 | tail -3 &&
for v in 1 2 3; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('RUN', d['ms_per_step'])" || exit 1; done
bash tools/run_r4_tl.sh > /dev/null 2>&1
