mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bf16b -o bf16 -- python $R/bench.py --dtype bf16 --batch 8 --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/prof_bf16b.log 2>&1
head -8 $R/gpurun_out/prof_bf16b/bf16_kernel_stats.csv | cut -c1-150
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bf16c -o p -- python $R/bench.py --dtype bf16 --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_bf16c.log 2>&1
tail -1 $R/gpurun_out/pmc_bf16c.log | cut -c1-300
