#!/bin/bash
# the whole GPU suite with the small-problem shapes disabled (256 x 128 / 128 x 256 everywhere, as in round 2) and forced
# (small shapes wherever they apply, full-size tests included): both shapes against every golden
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
CTTS_F32_NO_SMALL=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r3_sweep_no_small.log; tail -3 gpurun_out/r3_sweep_no_small.log
CTTS_F32_FORCE_SMALL=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r3_sweep_force_small.log; tail -3 gpurun_out/r3_sweep_force_small.log
