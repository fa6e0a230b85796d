cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "directional or f6" 2>&1 | tail -3
for v in 1 0 1 0; do GFDN_DIR_TWO_STREAMS=$v timeout 300 python bench.py --config directional --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two_streams=$v', d['config']['ms_per_band_step'])"; done
