#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_small_shape.py tests/test_gemm_mode.py -m gpu -q -x -s 2>&1 | grep -i "split\|passed\|failed\|error" | tail -8
echo "== split mode rows (library default bf16x3)"
CTTS_F32_GEMM_MODE=bf16x3 timeout 900 python scripts/bench_rows.py --rows waveglow_ax,waveglow_ax_untts,stft --steps 3 --warmup 1 2>/dev/null > gpurun_out/r3_z_rows_split.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r3_z_rows_split.jsonl"):
    d = json.loads(l); print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("row", "value", "batch", "ms_per_call", "f32_gemm_mode")})
PY
