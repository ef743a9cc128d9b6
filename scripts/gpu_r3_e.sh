#!/bin/bash
# all-gather floor microbenchmark; Tacotron tests + timeline + row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-e}
true
true
timeout 900 python -m pytest tests/test_tacotron.py tests/test_tacotron_long.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python scripts/profile_persistent.py > gpurun_out/r3_${tag}_persistent_timeline.txt 2>&1; tail -6 gpurun_out/r3_${tag}_persistent_timeline.txt
timeout 600 python scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows_tacotron.jsonl
cut -c1-200 gpurun_out/r3_${tag}_rows_tacotron.jsonl
