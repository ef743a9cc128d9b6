#!/bin/bash
# every secondary fp32 row with the library default set to the split loops (extra rows)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in bf16x3 bf16x6; do
  CTTS_F32_GEMM_MODE=$mode timeout 900 python scripts/bench_rows.py --rows waveflow,waveflow_author,waveglow_ax,waveglow_ax_untts --steps 3 --warmup 1 2>/dev/null > gpurun_out/r3_rows_split_$mode.jsonl
  python - "$mode" <<'PY'
import json, sys
for l in open(f"gpurun_out/r3_rows_split_{sys.argv[1]}.jsonl"):
    d = json.loads(l)
    print(sys.argv[1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("row", "value", "batch", "ms_per_call")})
PY
done
