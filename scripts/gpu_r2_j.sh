#!/bin/bash
# round 2: split-bf16 (bf16x3) path - parity vs the fp32 reference goldens, bf16 regression, then bench rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_waveglow_gpu.py tests/test_full_size.py -m gpu -x -q -k "bf16 or options or full_length" 2>&1 | tail -25 | tee gpurun_out/r2_j_pytest.log
for d in bf16x3 bf16; do
python bench.py --dtype $d --batch 8 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null > gpurun_out/r2_j_$d.json
python - <<PY
import json
d=json.load(open("gpurun_out/r2_j_$d.json")); r=d["roofline"]
print("$d", "ms/step", round(d["ms_per_step"],2), "value", round(d["value"]), "in-layer", r["mean_launch_ms"], r["frac"])
PY
done
