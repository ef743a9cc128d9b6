#!/bin/bash
# round 4, row queue: whole GPU suite, the config-4 rows, kernel stats and PMC of the B = 8 row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash scripts/gpu.sh tests r4_12
timeout 600 python scripts/bench_rows.py --rows waveflow --steps 5 --warmup 2 --batches 1,2,3,4,5,6,8 > gpurun_out/r4_12_rows_waveflow.jsonl 2> gpurun_out/r4_12_rows_waveflow.err
tail -3 gpurun_out/r4_12_rows_waveflow.jsonl | cut -c1-400
bash scripts/gpu.sh stats r4_12_waveflow_b8 scripts/bench_rows.py --rows waveflow --steps 3 --warmup 1 --batches 8
bash scripts/gpu.sh pmc waveflow_b8_queue scripts/bench_rows.py --rows waveflow --steps 2 --warmup 1 --batches 8
