cd $GRAFT_REPO_ROOT
python tools/tf8_probe.py 2>&1 | tail -7
timeout 600 python -m pytest tests/test_gpu_blocktf8.py -x -q 2>&1 | tail -3
