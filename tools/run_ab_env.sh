# same-box A/B of one environment switch on the headline bench:  bash tools/run_ab_env.sh VAR [bench args]
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in 1 0 1 0; do env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --steps 400 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'])"; done
