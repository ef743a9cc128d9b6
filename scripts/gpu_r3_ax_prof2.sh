#!/bin/bash
# ax notebook row at batch 1 after the flow-boundary fusion and the pipelined small conv-GEMM: kernel stats + PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_ax2_prof -o ax -- python3 $GRAFT_REPO_ROOT/scripts/bench_rows.py --rows waveglow_ax --batches 1 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_ax2_prof.log 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/r3_ax2_prof/ax_kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
bash scripts/pmc3.sh ax_b1_pipelined scripts/bench_rows.py --rows waveglow_ax --batches 1 --steps 1 --warmup 0
timeout 120 scripts/micro/bin/small_gemm_timeline > gpurun_out/r3_small_gemm_timeline_final.txt 2>&1
