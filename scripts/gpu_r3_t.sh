#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tacotron.py tests/test_tacotron_long.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 2>gpurun_out/r3_t_rows.err > gpurun_out/r3_t_rows_tacotron.jsonl
cut -c1-330 gpurun_out/r3_t_rows_tacotron.jsonl
