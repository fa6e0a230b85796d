cd $GRAFT_REPO_ROOT
cp diffgfdn_amd/bankstep.py /tmp/bankstep_new.py
for v in new old new old; do
  if [ $v = old ]; then cp tools/_bankstep_old.py diffgfdn_amd/bankstep.py; else cp /tmp/bankstep_new.py diffgfdn_amd/bankstep.py; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
done
cp /tmp/bankstep_new.py diffgfdn_amd/bankstep.py
timeout 600 python -m pytest tests/test_gpu_bank.py tests/test_gpu_fullsize.py -q -x -k "bank or bench_shape" 2>&1 | tail -2
