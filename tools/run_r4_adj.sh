#!/bin/bash
# round 4: adjoint STFT of the gradient spectra as one launch (odd frames to a second signal set): tests + same-box A/B
set -x -o pipefail
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -k "stft_pairs_spectrum or gamma_and_dots or composed" -x -q 2>&1 | tail -5 &&
timeout -k 10 900 python -m pytest tests/test_gpu_bank.py -x -q -k "oracle or stored or spectral or distinct" 2>&1 | tail -5 &&
for v in 0 1 0 1; do GFDN_ADJ_STFT_ONE_LAUNCH=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('AB adj_one_launch=$v', d['ms_per_step'])" || exit 1; done
