#!/bin/bash
# round 2: bf16 deferred skip GEMM - parity then the config 3 lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_waveglow_gpu.py tests/test_full_size.py -m gpu -x -q -k "bf16 or options" 2>&1 | tail -15 | tee gpurun_out/r2_g_pytest.log
python bench.py --dtype bf16 --batch 8 --steps 4 --warmup 1 --cpu-frames 0 2>/dev/null > gpurun_out/r2_g_bf16_b8.json
python bench.py --dtype bf16 --batch 32 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null > gpurun_out/r2_g_bf16_b32.json
python - <<PY
import json
for n in ("b8","b32"):
    d=json.load(open(f"gpurun_out/r2_g_bf16_{n}.json")); r=d["roofline"]
    print(n, "ms/step", round(d["ms_per_step"],2), "value", round(d["value"]), "in-layer", r["mean_launch_ms"], r["frac"],
          "res", r.get("res_hbm",{}).get("mean_launch_ms"), "skip", r.get("skip_hbm",{}).get("mean_launch_ms"))
PY
