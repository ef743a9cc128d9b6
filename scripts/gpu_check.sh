set -x; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r1_03.json 2> gpurun_out/bench_r1_03.err; tail -2 gpurun_out/bench_r1_03.err; cat gpurun_out/bench_r1_03.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch2 -o f -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_fetch2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l22 -o l -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_l22.log 2>&1
