#!/bin/bash
# small-problem shape of the fp32 conv-GEMM: bit-identity tests, the suites that now run through it, B=1 rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-s}
timeout 900 python -m pytest tests/test_small_shape.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -25
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_${tag}_pytest.log; tail -3 gpurun_out/r3_${tag}_pytest.log
timeout 900 python scripts/bench_rows.py --rows waveglow_ax,waveglow_ax_untts,waveflow_author,tacotron,stft --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows.jsonl
python - "$tag" <<'PY'
import json
import sys
for l in open(f"gpurun_out/r3_{sys.argv[1]}_rows.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("row", "value", "batch", "ms_per_call", "end_to_end_ms_incl_encoder_postnet", "achieved_tflops_wn_gemms")})
PY
