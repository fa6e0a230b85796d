cd $GRAFT_REPO_ROOT
timeout 300 python tools/dir_ops_probe.py > gpurun_out/dir_ops.txt 2>&1; tail -5 gpurun_out/dir_ops.txt
