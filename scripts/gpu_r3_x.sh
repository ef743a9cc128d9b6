#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/bench_rows.py --rows waveflow --batches 1,2,4,8 --steps 3 --warmup 1 2>/dev/null > gpurun_out/r3_x_rows_waveflow_batches.jsonl
CTTS_F32_NO_SMALL=1 timeout 600 python scripts/bench_rows.py --rows waveflow --batches 1,2,4 --steps 3 --warmup 1 2>/dev/null | sed 's/"row": "B\/config4"/"row": "B\/config4 (CTTS_F32_NO_SMALL=1: 128x256 blocks)"/' >> gpurun_out/r3_x_rows_waveflow_batches.jsonl
cut -c1-120 gpurun_out/r3_x_rows_waveflow_batches.jsonl
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
