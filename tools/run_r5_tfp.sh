#!/bin/bash
# round 5: the transform-based 8-line passes -- kernel tests, the N = 32 step tests, A/B and timeline
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_polyfft.py tests/test_gpu_blocktf8.py tests/test_gpu_fullsize.py -q -s -k "polyfft or blocktf8 or transform_passes or n32 or shape_vs_oracle" > $OUT/r5_tfp_tests.log 2>&1; echo "tests rc=$?"; grep "parity\|passed\|failed" $OUT/r5_tfp_tests.log | cut -c1-160
B="--lines-per-group 8 --no-cpu-baseline --no-extras --steps 300"
for i in 1 2; do
  timeout -k 10 300 python tools/ab_attr.py -- $B 2>/dev/null | tail -1
  timeout -k 10 300 python tools/ab_attr.py transform_polys=False -- $B 2>/dev/null | tail -1
done | tee $OUT/r5_tfp_ab.log
rm -rf $OUT/n32_stats; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/n32_stats -- python $GRAFT_REPO_ROOT/bench.py $B > $OUT/n32_stats.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/n32_stats 60 k_tf8_coefs > $OUT/r5_tfp_timeline.txt 2>&1; tail -45 $OUT/r5_tfp_timeline.txt
