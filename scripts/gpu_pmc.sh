set -x; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2 -o l -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_l2.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_l2; tail -2 $R/gpurun_out/pmc_l2.log
