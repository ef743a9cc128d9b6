#!/bin/bash
# ax notebook row at batch 8 on the small shape (threshold 2048): counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash scripts/pmc3.sh ax_b8_small scripts/bench_rows.py --rows waveglow_ax --batches 8 --steps 1 --warmup 0
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3_pmc_ax_b8_small.json"))
for k, v in d["kernels"].items():
    if v.get("mean_us_under_pmc", 0) > 20:
        print(k[:60], {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("dispatches_per_pass", "mean_us_under_pmc", "mfma_busy_frac_per_simd", "FETCH_SIZE", "WRITE_SIZE", "vgpr") if x in v})
PY
