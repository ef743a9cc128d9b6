#!/bin/bash
# bf16x6 (three-way split, six products): parity tests, bench line next to bf16x3 and fp32, ax / waveflow rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-x6}
timeout 1500 python -m pytest tests/test_gemm_mode.py tests/test_small_shape.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -22 > gpurun_out/r3_${tag}_pytest.log; cat gpurun_out/r3_${tag}_pytest.log
for mode in bf16x6 bf16x3; do
python bench.py --steps 5 --warmup 2 --gemm-mode $mode --cpu-frames 0 --no-exchange-dry-run > gpurun_out/r3_${tag}_bench_f32_split_$mode.json 2>/dev/null
python - "$mode" "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r3_{sys.argv[2]}_bench_f32_split_{sys.argv[1]}.json") if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["frac"], d["roofline"]["mean_launch_ms"])
PY
done
CTTS_F32_GEMM_MODE=bf16x6 timeout 900 python scripts/bench_rows.py --rows waveflow,waveglow_ax,waveglow_ax_untts --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows_bf16x6.jsonl
python - "$tag" <<'PY'
import json
import sys
for l in open(f"gpurun_out/r3_{sys.argv[1]}_rows_bf16x6.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("row", "value", "batch", "ms_per_call")})
PY
