#!/bin/bash
# the whole GPU suite, then a short bench
: "${GRAFT_REPO_ROOT:?}"
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -u -m pytest tests -q -m gpu -x > $OUT/r5_suite.log 2>&1; rc=$?; tail -25 $OUT/r5_suite.log
[ $rc -ne 0 ] && exit 1
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'])"; done
