# directional configuration: parity tests of its kernels and routes + the band-step time (one gpurun call)
cd $GRAFT_REPO_ROOT
set -o pipefail
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -k "pow2 or directional or compose_sh or sh_ or edc or normalise or f6 or solve" 2>&1 | tail -3 && \
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k directional 2>&1 | tail -3 && \
timeout 300 python bench.py --config directional --no-cpu-baseline > gpurun_out/dir_bench.json 2> gpurun_out/dir_bench.err && python - <<'PY'
import json
d = json.loads(open('gpurun_out/dir_bench.json').read().strip().splitlines()[-1])
print('ms_per_band_step', d['config']['ms_per_band_step'], 'inverse transform pair us', d.get('roofline', {}).get('avg_launch_us'))
PY
