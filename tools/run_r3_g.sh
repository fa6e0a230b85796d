cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "f6 or directional" 2>&1 | tail -4
timeout 300 python bench.py --config directional --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('directional ms per band-step', d['config']['ms_per_band_step'], d['config']['final_loss'])"
