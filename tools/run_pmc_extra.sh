set -x
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r01_pmc_tcc gpurun_out/r01_pmc_sq
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_tcc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_tcc/*/ $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_sq/*/ | head
