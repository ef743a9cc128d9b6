#!/bin/bash
# round 2: ax-core WaveGlow (waveflow=False) parity + the notebook-config row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_waveglow_ax.py tests/test_waveflow.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2_b_pytest.log
tail -12 gpurun_out/r2_b_pytest.log
python scripts/bench_rows.py --rows waveglow_ax --steps 3 --warmup 1 > gpurun_out/r2_b_rows.jsonl 2> gpurun_out/r2_b_rows.err
cat gpurun_out/r2_b_rows.jsonl; tail -5 gpurun_out/r2_b_rows.err
cd /tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2_b_prof" -o ax -- python "$GRAFT_REPO_ROOT/scripts/bench_rows.py" --rows waveglow_ax --steps 1 --warmup 1 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find gpurun_out/r2_b_prof -name "*kernel_stats.csv" | head -1); head -20 "$f"
