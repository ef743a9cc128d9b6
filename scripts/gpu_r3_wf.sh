#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-wf}
timeout 1200 python -m pytest tests/test_waveflow.py tests/test_small_shape.py tests/test_full_size.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r3_${tag}_pytest.log; tail -3 gpurun_out/r3_${tag}_pytest.log
timeout 900 python scripts/bench_rows.py --rows waveflow --batches 1,2,3,4,8 --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows.jsonl
python - "$tag" <<'PY'
import json
import sys
for l in open(f"gpurun_out/r3_{sys.argv[1]}_rows.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("row", "value", "batch", "ms_per_call")})
PY
