#!/bin/bash
# counters of the headline launch (own --pmc passes): MFMA busy, shader clock, traffic
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-headline}
bash scripts/pmc3.sh $tag bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-cpu-aggregate --no-exchange-dry-run --no-kernel-timing
python3 - "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3_pmc_{sys.argv[1]}.json"))
for k, v in d["kernels"].items():
    if "conv_gemm_f32_kernel" in k:
        print(k, {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("mean_us_under_pmc", "shader_clock_ghz_under_pmc", "mfma_busy_frac_per_simd", "FETCH_SIZE", "WRITE_SIZE", "vgpr") if x in v})
PY
