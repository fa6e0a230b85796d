# directional baseline: kernel table + torch-operator probe
cd $GRAFT_REPO_ROOT
bash tools/run_dir_profile.sh > gpurun_out/dir_kernels.txt 2>&1 && timeout 300 python tools/dir_ops_probe.py > gpurun_out/dir_ops.txt 2>&1
tail -5 gpurun_out/dir_kernels.txt
