cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/k_head.json 2> gpurun_out/k_head.err && \
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
