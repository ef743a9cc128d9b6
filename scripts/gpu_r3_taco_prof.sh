#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_taco_prof2 -o taco -- python3 $GRAFT_REPO_ROOT/scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_taco_prof2.log 2>&1
head -14 $GRAFT_REPO_ROOT/gpurun_out/r3_taco_prof2/taco_kernel_stats.csv | cut -c1-150
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r3_taco_prof2.log | cut -c1-300
