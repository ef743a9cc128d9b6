#!/bin/bash
# res/skip GEMM of the headline model in the small shape (forced): per-launch time next to the large shape's 1.89 ms
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export CTTS_F32_FORCE_SMALL=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_rs_small_prof -o f32 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-exchange-dry-run > $GRAFT_REPO_ROOT/gpurun_out/r3_rs_small_prof.log 2>&1
head -4 $GRAFT_REPO_ROOT/gpurun_out/r3_rs_small_prof/f32_kernel_stats.csv | cut -c1-170
