set -x
cd $GRAFT_REPO_ROOT
timeout 200 python tools/grad_stage_probe.py 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_decay.py tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8 && \
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s 2>&1 | grep -E "passed|failed|deviation|Error|error" | tail -15 && \
python tools/decay_probe.py 224 && \
python bench.py --no-cpu-baseline --steps 400 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['build'])"
