# the driver's GPU test command, twice in a row, with full logs (flakiness check)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout -k 10 560 python -m pytest tests -x -q -m gpu > gpurun_out/suite_rep$i.log 2>&1; echo "run $i rc=$?"
  grep -n "Fatal Python\|Segmentation\|Aborted\|passed\|failed" gpurun_out/suite_rep$i.log | head -5
done
