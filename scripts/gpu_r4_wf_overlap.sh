#!/bin/bash
# upper bound of what overlapping the layers of a WaveFlow row could win (profiles/r4_10_waveflow_overlap_bound.txt).
# The library given to this script is built with -DCTTS_WF_OVERLAP_EXPERIMENT (results garbage, timing only).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do timeout 300 python scripts/bench_rows.py --rows waveflow --steps 3 2>/dev/null | cut -c1-330; done
