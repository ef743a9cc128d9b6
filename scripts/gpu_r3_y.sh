#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_y_prof -o wf -- python3 $GRAFT_REPO_ROOT/scripts/bench_rows.py --rows waveflow --batches 1 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_y_prof.log 2>&1
head -12 $GRAFT_REPO_ROOT/gpurun_out/r3_y_prof/wf_kernel_stats.csv | cut -c1-200
python3 - <<'PY'
import csv, os
p = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3_y_prof/wf_kernel_trace.csv"
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last call: take the last 1/5 of the dispatches
n = len(rows) // 5
last = rows[-n:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print(f"last call: {n} dispatches, kernel time {busy/1e6:.2f} ms of a {span/1e6:.2f} ms span -> gaps {100*(1-busy/span):.1f} %")
PY
