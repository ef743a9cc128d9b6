#!/bin/bash
# whole GPU suite, then every secondary row and the small-launch timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-all}
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_${tag}_pytest.log; tail -4 gpurun_out/r3_${tag}_pytest.log
timeout 1200 python scripts/bench_rows.py --rows waveflow,waveflow_author,waveglow_ax,waveglow_ax_untts,tacotron,stft --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows.jsonl
python - "$tag" <<'PY'
import json
import sys
for l in open(f"gpurun_out/r3_{sys.argv[1]}_rows.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("row", "value", "batch", "ms_per_call", "end_to_end_ms_incl_encoder_postnet")})
PY
timeout 120 scripts/micro/bin/small_gemm_timeline > gpurun_out/r3_${tag}_small_gemm_timeline.txt 2>&1; grep "blocks,\|tables\|first chunk" gpurun_out/r3_${tag}_small_gemm_timeline.txt
