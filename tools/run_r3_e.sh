set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_blocktf8.py -x -q 2>&1 | tail -25
