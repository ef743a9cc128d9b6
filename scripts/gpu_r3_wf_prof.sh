#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_wf_prof -o wf -- python3 $GRAFT_REPO_ROOT/scripts/bench_rows.py --rows waveflow --batches 1 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_wf_prof.log 2>&1
head -14 $GRAFT_REPO_ROOT/gpurun_out/r3_wf_prof/wf_kernel_stats.csv | cut -c1-170
