#!/bin/bash
# Tacotron loop: both test files, persistent-decoder phase timeline, the config-5 row, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-c}
timeout 900 python -m pytest tests/test_tacotron.py tests/test_tacotron_long.py tests/test_full_size.py -k "tacotron or config5" -m gpu -q -s -x 2>&1 | grep -v "^$" | tail -120 > gpurun_out/r3_${tag}_taco_tests.log
grep -c "^\." gpurun_out/r3_${tag}_taco_tests.log; tail -4 gpurun_out/r3_${tag}_taco_tests.log
timeout 300 python scripts/profile_persistent.py > gpurun_out/r3_${tag}_persistent_timeline.txt 2>&1; tail -7 gpurun_out/r3_${tag}_persistent_timeline.txt
timeout 600 python scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows_tacotron.jsonl
cut -c1-400 gpurun_out/r3_${tag}_rows_tacotron.jsonl
