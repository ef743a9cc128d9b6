#!/bin/bash
# counters of the author's WaveFlow row after the rewrite of the fused separable layer (own --pmc passes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash scripts/pmc3.sh waveflow_author_r3b scripts/bench_rows.py --rows waveflow_author --steps 1 --warmup 0
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3_pmc_waveflow_author_r3b.json"))
for k, v in d["kernels"].items():
    if v.get("mean_us_under_pmc", 0) > 20:
        print(k[:60], {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("dispatches_per_pass", "mean_us_under_pmc", "mfma_busy_frac_per_simd", "FETCH_SIZE", "WRITE_SIZE", "vgpr") if x in v})
PY
