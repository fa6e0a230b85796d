# round 4: the bench line end to end (default run with the two sub-records and the CPU baseline), the other bench legs, a
# 2-rank rehearsal on one device over gloo (strong headline + weak + band-sharded in one line)
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
( time timeout 600 python bench.py > $OUT/r04_bench_default.json 2> $OUT/r04_bench_default.err ) 2>&1 | tail -3; tail -c 600 $OUT/r04_bench_default.err; python - <<PY
import json
d=json.load(open('$OUT/r04_bench_default.json'))
print('ms', d['ms_per_step'], 'value', d['value'], 'scaling', d['scaling'])
print('roofline', {k: d['roofline'][k] for k in ('kernel','avg_launch_us','frac','isolated_us','traffic')})
print('extra', {k: (v['ms_per_step'], v.get('wall_s_incl_build')) for k, v in d.get('extra', {}).items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['loss_delta_vs_cpu']['total'], {k: v['max_dev_rel'] for k, v in d['cpu_baseline']['loss_delta_vs_cpu']['gradients'].items()})
PY
GFDN_BENCH_ONE_DEVICE=1 GFDN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/r04_b_w2.json 2> $OUT/r04_b_w2.err; tail -c 400 $OUT/r04_b_w2.err; python - <<PY
import json
d=json.load(open('$OUT/r04_b_w2.json'))
print('N=2 rehearsal (one device, gloo): headline', d['scaling'], d['ms_per_step'], d['value'], 'launch', d['config']['launch'])
print({k: (v['ms_per_step'], v['value']) for k, v in d['extra'].items()})
PY
timeout 300 python bench.py --epoch --steps 2 --warmup 1 --no-cpu-baseline > $OUT/r04_b_epoch.json 2> $OUT/r04_b_epoch.err; tail -c 300 $OUT/r04_b_epoch.err; python -c "import json;d=json.load(open('$OUT/r04_b_epoch.json'));print('epoch', d['epoch']['s_per_epoch_all_bands'], d['ms_per_step'])"
timeout 300 python bench.py --bands 1 --no-cpu-baseline > $OUT/r04_b_band1.json 2> $OUT/r04_b_band1.err; python -c "import json;d=json.load(open('$OUT/r04_b_band1.json'));print('1 band', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --distinct-t60 --no-cpu-baseline --no-extras > $OUT/r04_b_distinct.json 2> $OUT/r04_b_distinct.err; python -c "import json;d=json.load(open('$OUT/r04_b_distinct.json'));print('distinct t60', d['ms_per_step'], d['value'])"
