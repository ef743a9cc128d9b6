#!/bin/bash
# full GPU suite on the tree with the ragged-accumulate fix and the faster persistent decoder; timeline; config-5 row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_d_pytest.log
tail -3 gpurun_out/r3_d_pytest.log
timeout 600 python -m pytest tests/test_tacotron_long.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -80 > gpurun_out/r3_d_taco_long.log
tail -2 gpurun_out/r3_d_taco_long.log
timeout 300 python scripts/profile_persistent.py > gpurun_out/r3_d_persistent_timeline.txt 2>&1; tail -7 gpurun_out/r3_d_persistent_timeline.txt
timeout 600 python scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 2>gpurun_out/r3_d_rows.err > gpurun_out/r3_d_rows_tacotron.jsonl
cut -c1-300 gpurun_out/r3_d_rows_tacotron.jsonl
