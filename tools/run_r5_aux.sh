#!/bin/bash
# the measurement legs beside the headline (one gpurun call): directional records, bench checks, N = 32 kernel totals
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
bash tools/run_dir_measurements.sh r05 > gpurun_out/r05_dir_run.log 2>&1; echo "dir rc=$?"
bash tools/run_bench_checks.sh > gpurun_out/r05_checks_run.log 2>&1; echo "checks rc=$?"
bash tools/run_n32_profile.sh > gpurun_out/r05_n32_kernels.txt 2>&1; echo "n32 rc=$?"; head -12 gpurun_out/r05_n32_kernels.txt
python tools/timeline.py gpurun_out/n32_stats 30 k_tf8_coefs > gpurun_out/r05_n32_timeline.txt 2>&1; echo "n32 timeline rc=$?"
