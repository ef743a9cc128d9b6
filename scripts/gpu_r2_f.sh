#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in default CTTS_BF16_NO_WIDE CTTS_BF16_NO_PP; do
  if [ "$v" = default ]; then python bench.py --dtype bf16 --batch 8 --steps 4 --warmup 1 --cpu-frames 0 2>/dev/null > gpurun_out/r2_f_$v.json
  else env $v=1 python bench.py --dtype bf16 --batch 8 --steps 4 --warmup 1 --cpu-frames 0 2>/dev/null > gpurun_out/r2_f_$v.json; fi
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_f_$v.json"))
r=d["roofline"]
print("$v", "ms/step", round(d["ms_per_step"],2), "in-layer ms", r["mean_launch_ms"], "res/skip ms", r["res_skip_hbm"]["mean_launch_ms"])
PY
done
