#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_waveglow_ax.py tests/test_small_shape.py -m gpu -q -x -s 2>&1 | grep -i "c96\|c160\|passed\|failed\|error" | tail -12
