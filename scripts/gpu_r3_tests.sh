#!/bin/bash
# the whole GPU suite; tail to gpurun_out/r3_<tag>_pytest.log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-t}
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_${tag}_pytest.log; tail -6 gpurun_out/r3_${tag}_pytest.log
