#!/bin/bash
# the measurement legs beside the headline (one gpurun call):  bash tools/run_aux.sh [tag]
# directional records, bench checks (epoch, N = 32, distinct windows, one band, 2-rank gloo rehearsals), N = 32 kernel totals
# and timeline, the reference recipe's line and timeline
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/run_dir_measurements.sh $TAG > gpurun_out/${TAG}_dir_run.log 2>&1; echo "dir rc=$?"
bash tools/run_bench_checks.sh > gpurun_out/${TAG}_checks_run.log 2>&1; echo "checks rc=$?"
bash tools/run_n32_profile.sh > gpurun_out/${TAG}_n32_kernels.txt 2>&1; echo "n32 rc=$?"; head -12 gpurun_out/${TAG}_n32_kernels.txt
python tools/timeline.py gpurun_out/n32_stats 30 k_tf8_coefs > gpurun_out/${TAG}_n32_timeline.txt 2>&1; echo "n32 timeline rc=$?"
timeout 300 python bench.py --recipe reference --no-cpu-baseline > gpurun_out/${TAG}_recipe_bench.json 2> gpurun_out/${TAG}_recipe_bench.err; echo "recipe rc=$?"
bash tools/run_timeline.sh ${TAG}_recipe --recipe reference > /dev/null 2>&1; echo "recipe timeline rc=$?"
