#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2_d_pytest.log
tail -8 gpurun_out/r2_d_pytest.log
python scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 2>/dev/null | cut -c1-600
