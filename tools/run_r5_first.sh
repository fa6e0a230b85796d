#!/bin/bash
# round 5, first GPU call: new-kernel tests, same-box A/B of the three switches, timeline of the replayed step
: "${GRAFT_REPO_ROOT:?}"
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
echo skip > $OUT/r5_tests_a.log; rc=0
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "round5 or fused_tail_steps" 2>&1 | tail -15 > $OUT/r5_tests_b.log; rc=$?; cat $OUT/r5_tests_b.log
[ $rc -ne 0 ] && exit 1
ab() {
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"
}
abx() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"
}
export GFDN_COLORLESS_TAIL=0
for i in 1 2; do
  abx --chain-steps 1
  abx --chain-steps 2
  abx --chain-steps 4
  abx --chain-steps 8
done 2>&1 | tee $OUT/r5_ab.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 --bands 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bands=1', d['ms_per_step'])" | tee -a $OUT/r5_ab.log
rm -rf $OUT/r05tl_stats
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05tl_stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --chain-steps 4 > $OUT/r05tl_stats.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/r05tl_stats 100 > $OUT/r05_timeline.txt 2>&1; tail -45 $OUT/r05_timeline.txt
