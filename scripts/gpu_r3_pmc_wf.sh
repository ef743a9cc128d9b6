#!/bin/bash
# counters of the split-K fused WaveFlow layer (config 4 at batch 1): own --pmc passes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash scripts/pmc3.sh waveflow_b1_splitk scripts/bench_rows.py --rows waveflow --batches 1 --steps 1 --warmup 0
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3_pmc_waveflow_b1_splitk.json"))
for k, v in d["kernels"].items():
    print(k[:70], {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("dispatches_per_pass", "mean_us_under_pmc", "shader_clock_ghz_under_pmc", "mfma_busy_frac_per_simd", "FETCH_SIZE", "WRITE_SIZE", "vgpr", "lds") if x in v})
PY
