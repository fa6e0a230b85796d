# bench legs beside the headline line (one gpurun call): whole epochs, N = 32 bank, directional, 2-rank rehearsals on one
# device over gloo (weak and strong)
set -x
cd $GRAFT_REPO_ROOT
true && \
timeout 300 python bench.py --epoch --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_epoch.json 2> gpurun_out/b_epoch.err && tail -c 900 gpurun_out/b_epoch.json && \
timeout 300 python bench.py --lines-per-group 8 --no-cpu-baseline > gpurun_out/b_n32.json 2> gpurun_out/b_n32.err && tail -c 400 gpurun_out/b_n32.json && \
timeout 300 python bench.py --distinct-t60 --no-cpu-baseline --no-extras > gpurun_out/b_t60.json 2> gpurun_out/b_t60.err && tail -c 300 gpurun_out/b_t60.json && \
timeout 300 python bench.py --bands 1 --no-cpu-baseline > gpurun_out/b_band1.json 2> gpurun_out/b_band1.err && tail -c 300 gpurun_out/b_band1.json && \
timeout 300 python bench.py --config directional --no-cpu-baseline > gpurun_out/b_dir.json 2> gpurun_out/b_dir.err && tail -c 400 gpurun_out/b_dir.json && \
GFDN_BENCH_ONE_DEVICE=1 GFDN_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --extra-steps 10 > gpurun_out/b_w2.json 2> gpurun_out/b_w2.err && tail -c 600 gpurun_out/b_w2.json && \
GFDN_BENCH_ONE_DEVICE=1 GFDN_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --scaling weak > gpurun_out/b_w2s.json 2> gpurun_out/b_w2s.err && tail -c 600 gpurun_out/b_w2s.json
