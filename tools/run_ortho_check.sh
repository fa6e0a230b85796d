cd $GRAFT_REPO_ROOT
set -o pipefail
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_bank.py tests/test_gpu_blocktf.py tests/test_gpu_blocktf8.py -q -x -k "ortho or bank or tail or param or f1 or f4 or directional" 2>&1 | tail -3 && \
timeout 300 python bench.py --no-cpu-baseline --steps 400 > gpurun_out/k_head.json 2> gpurun_out/k_head.err && \
timeout 300 python bench.py --lines-per-group 8 --no-cpu-baseline --steps 300 > gpurun_out/k_n32.json 2> gpurun_out/k_n32.err && \
timeout 300 python bench.py --config directional --no-cpu-baseline > gpurun_out/k_dir.json 2> gpurun_out/k_dir.err; python - <<'PY'
import json
for f in ('k_head', 'k_n32', 'k_dir'):
    try:
        d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['config'].get('ms_per_band_step'))
    except Exception as e:
        print(f, 'failed', e)
PY
