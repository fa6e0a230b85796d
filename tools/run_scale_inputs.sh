#!/bin/bash
# step times a rank of an N-GPU job would see (one device): inputs of DESIGN.md section 7's expected-scaling table
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],4))"; }
run --bands 7 --batch 32
run --bands 7 --batch 16
run --bands 7 --batch 8
run --bands 7 --batch 4
run --bands 1 --batch 32
run --bands 1 --batch 4
run --bands 4 --batch 32
run --bands 2 --batch 32
