#!/bin/bash
# round 4: trainer.DirectionalBank (one graph for all bands): test + same-box A/B of the directional bench
set -x -o pipefail
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "directional" -x -q 2>&1 | tail -5 &&
for v in "0 2" "1 2" "1 4" "0 2" "1 4"; do set -- $v; timeout -k 10 300 python bench.py --config directional --no-cpu-baseline --steps 40 --dir-bank $1 --dir-streams $2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('AB dir_bank=$1 lanes=$2', d['ms_per_step'], d['config']['ms_per_band_step'])" || exit 1; done
