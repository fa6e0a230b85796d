#!/bin/bash
# round 4: EDR loss with the receivers walked by cell-owning threads: kernel test, bank tests, same-box A/B of the launch's forms
# (GFDN_EDR_FORM: 0 = k_edr_lin_band, LDS exchanges + two barriers per receiver; 1 = k_edr_lin_wave, scans inside the wave)
set -x -o pipefail
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -k "composed_spectra or stft_pairs_spectrum" -x -q 2>&1 | tail -5 &&
timeout -k 10 900 python -m pytest tests/test_gpu_bank.py -x -q -k "oracle or stored or spectral" 2>&1 | tail -5 &&
for v in "0 0" "1 0" "0 0" "1 0"; do set -- $v; GFDN_EDR_FORM=$1 GFDN_EDR_RUNS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('AB form=$1 runs=$2', d['ms_per_step'], d.get('roofline', {}).get('avg_launch_us'), d.get('roofline', {}).get('isolated_us'))" || exit 1; done
