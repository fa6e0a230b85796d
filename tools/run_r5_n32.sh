#!/bin/bash
# round 5: N = 32 bank -- tests of the 8-line path, A/B of the colorless pass's position, timeline
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_blocktf8.py tests/test_gpu_bank.py -x -q -k "eight or tf8 or blocktf8 or fused_bank_step" > $OUT/r5_n32_tests.log 2>&1; tail -3 $OUT/r5_n32_tests.log
B="--lines-per-group 8 --no-cpu-baseline --no-extras --steps 300"
for i in 1 2; do
  timeout -k 10 300 python tools/ab_attr.py -- $B 2>/dev/null | tail -1
  timeout -k 10 300 python tools/ab_attr.py colorless_behind_scans=False -- $B 2>/dev/null | tail -1
done | tee $OUT/r5_n32_ab.log
rm -rf $OUT/n32_stats; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/n32_stats -- python $GRAFT_REPO_ROOT/bench.py $B > $OUT/n32_stats.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/n32_stats 60 k_tf8_coefs | tail -40
