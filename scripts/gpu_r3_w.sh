#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_small_shape.py tests/test_waveflow.py tests/test_full_size.py -k "waveflow or config4" -m gpu -q -x -s 2>&1 | grep -i "large shape\|passed\|failed\|error\|config 4" | tail -12
echo "== default (small below 256 blocks)"
timeout 600 python scripts/bench_rows.py --rows waveflow --batches 1,2,4,8 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','batch','ms_per_call','rtf')})"
echo "== forced small"
CTTS_F32_FORCE_SMALL=1 timeout 600 python scripts/bench_rows.py --rows waveflow --batches 4,8 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','batch','ms_per_call','rtf')})"
echo "== never small"
CTTS_F32_NO_SMALL=1 timeout 600 python scripts/bench_rows.py --rows waveflow --batches 1,4 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','batch','ms_per_call','rtf')})"
