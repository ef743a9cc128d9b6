set -x; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r1_02.json 2> gpurun_out/bench_r1_02.err; tail -2 gpurun_out/bench_r1_02.err; cat gpurun_out/bench_r1_02.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_02 -o p02 -- python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/prof_02.log 2>&1
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_02a -o a -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_02a.log 2>&1
tail -2 $R/gpurun_out/pmc_02a.log
ls -R $R/gpurun_out/prof_02 $R/gpurun_out/pmc_02a | head -20
