#!/bin/bash
# counters of the config-2 in-layer launch under the six-product loop (own --pmc passes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash scripts/pmc3.sh config2_bf16x6 bench.py --steps 1 --warmup 1 --gemm-mode bf16x6 --cpu-budget 0 --cpu-frames 0 --no-cpu-aggregate --no-exchange-dry-run --no-kernel-timing
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3_pmc_config2_bf16x6.json"))
for k, v in d["kernels"].items():
    if "conv_gemm_f32_kernel" in k:
        print(k, {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("mean_us_under_pmc", "shader_clock_ghz_under_pmc", "mfma_busy_frac_per_simd", "FETCH_SIZE", "WRITE_SIZE", "vgpr") if x in v})
PY
