#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2; do
  timeout -k 10 300 python tools/ab_attr.py -- --lines-per-group 8 --no-cpu-baseline --no-extras --steps 300 2>/dev/null | tail -1
  timeout -k 10 300 python tools/ab_attr.py -- --no-cpu-baseline --no-extras --steps 400 2>/dev/null | tail -1
  timeout -k 10 300 python tools/ab_attr.py -- --no-cpu-baseline --no-extras --steps 400 --bands 1 2>/dev/null | tail -1
done | tee $OUT/r5_prio.log
rm -rf $OUT/n32_stats; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/n32_stats -- python $GRAFT_REPO_ROOT/bench.py --lines-per-group 8 --no-cpu-baseline --no-extras --steps 200 > $OUT/n32_stats.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/n32_stats 60 k_tf8_coefs | grep "q=  1" | tail -24
