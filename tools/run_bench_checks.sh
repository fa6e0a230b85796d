set -x
cd $GRAFT_REPO_ROOT
true && \
timeout 300 python bench.py --epoch --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_epoch.json 2> gpurun_out/b_epoch.err && tail -c 1500 gpurun_out/b_epoch.json && \
GFDN_BENCH_ONE_DEVICE=1 GFDN_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b_w2.json 2> gpurun_out/b_w2.err && tail -c 1200 gpurun_out/b_w2.json && \
GFDN_BENCH_ONE_DEVICE=1 GFDN_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --scaling strong > gpurun_out/b_w2s.json 2> gpurun_out/b_w2s.err && tail -c 600 gpurun_out/b_w2s.json
