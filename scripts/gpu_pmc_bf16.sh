mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bf16 -o p -- python $R/bench.py --dtype bf16 --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_bf16.log 2>&1

# (FETCH_SIZE together with TCC_HIT_sum / TCC_MISS_sum in one pass did not finish within 1200 s on this pool: collect them separately)
