#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
CTTS_F32_FORCE_SMALL=1 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-exchange-dry-run > gpurun_out/r3_force_small_bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_force_small_bench.json") if l.startswith("{")][-1])
print("FORCE_SMALL headline:", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["mean_launch_ms"])
PY
CTTS_F32_FORCE_SMALL=1 timeout 600 python scripts/bench_rows.py --rows waveglow_ax --batches 8 --steps 2 --warmup 1 2>/dev/null | cut -c1-250
