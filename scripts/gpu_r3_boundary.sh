#!/bin/bash
# fused flow boundary of the ax 1-D core (end 1x1 + coupling + un-mix + start in one launch): parity, rows, and the
# per-phase timeline of the short conv-GEMM launches (scripts/micro/small_gemm_timeline.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-boundary}
timeout 1200 python -m pytest tests/test_waveglow_ax.py tests/test_small_shape.py tests/test_full_size.py tests/test_gemm_mode.py tests/test_conv1d_primitive.py tests/test_tacotron.py tests/test_stft.py tests/test_waveflow.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r3_${tag}_pytest_ax.log; tail -3 gpurun_out/r3_${tag}_pytest_ax.log
timeout 900 python scripts/bench_rows.py --rows waveglow_ax,waveglow_ax_untts,waveflow,tacotron,stft --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows.jsonl
python - "$tag" <<'PY'
import json
import sys
for l in open(f"gpurun_out/r3_{sys.argv[1]}_rows.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("row", "value", "batch", "ms_per_call", "end_to_end_ms_incl_encoder_postnet")})
PY
for v in ""; do
  [ -x scripts/micro/bin/small_gemm_timeline$v ] || continue
  echo "== variant: ${v:-product}" >> gpurun_out/r3_${tag}_small_gemm_timeline.txt
  timeout 120 scripts/micro/bin/small_gemm_timeline$v >> gpurun_out/r3_${tag}_small_gemm_timeline.txt 2>&1
done
cat gpurun_out/r3_${tag}_small_gemm_timeline.txt
