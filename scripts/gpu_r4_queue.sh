#!/bin/bash
# row queue: tests, then config-4 rows at several batch sizes: default / queue forced / queue off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_waveflow_row_queue.py -m gpu -q -s -x 2>&1 | tail -40 > gpurun_out/r4_q_pytest.log
tail -5 gpurun_out/r4_q_pytest.log
grep -q "passed" gpurun_out/r4_q_pytest.log || exit 1
for mode in default forced off; do
  case $mode in
    default) envs="";;
    forced) envs="CTTS_WF_ROW_QUEUE_MIN=1";;
    off) envs="CTTS_WF_NO_ROW_QUEUE=1";;
  esac
  env $envs timeout 600 python scripts/bench_rows.py --rows waveflow --steps 5 --warmup 2 --batches 1,2,3,4,8 > gpurun_out/r4_q_rows_$mode.jsonl 2> gpurun_out/r4_q_rows_$mode.err
  python - <<P
import json
for l in open("gpurun_out/r4_q_rows_$mode.jsonl"):
    l=l.strip()
    if l.startswith("{"):
        r=json.loads(l); print("$mode", r["batch"], round(r["ms_per_call"],2), "ms", round(r["mfma_frac_algorithmic"],3), r["last_gemm_loop"])
P
done
