#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_wfa_prof -o wfa -- python3 $GRAFT_REPO_ROOT/scripts/bench_rows.py --rows waveflow_author --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_wfa_prof.log 2>&1
head -16 $GRAFT_REPO_ROOT/gpurun_out/r3_wfa_prof/wfa_kernel_stats.csv | cut -c1-190
