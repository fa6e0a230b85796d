#!/bin/bash
# round 4: the colorless pass on a stream of its own (GFDN_COLORLESS_STREAM): bank tests + same-box A/B at N = 16 and N = 32
set -x -o pipefail
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_bank.py -x -q 2>&1 | tail -3 &&
for v in 0 1 0 1; do GFDN_COLORLESS_STREAM=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('AB16 own_stream=$v', d['ms_per_step'])" || exit 1; done
for v in 0 1 0 1; do GFDN_COLORLESS_STREAM=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --lines-per-group 8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('AB32 own_stream=$v', d['ms_per_step'])" || exit 1; done
