#!/bin/bash
# round 4: one stats run of the N = 32 bench for a timeline of the replayed step
set -x -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/r04tl32_stats
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04tl32_stats -- python $GRAFT_REPO_ROOT/bench.py --lines-per-group 8 --no-cpu-baseline --no-extras --steps 150 > $OUT/r04tl32_stats.log 2>&1
tail -c 300 $OUT/r04tl32_stats.log
