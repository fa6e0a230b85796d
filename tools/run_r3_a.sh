# round 3, first call: the fused decay kernel's tests, the bank / full-size tests on top of it, and a same-box A/B
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_decay.py -x -q 2>&1 | tail -15 && \
timeout 600 python -m pytest tests/test_gpu_bank.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 && \
bash tools/ab_env.sh GFDN_FUSE_DECAY 2
