#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/bench_rows.py --rows waveflow --batches 1,2,8 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('row','value','batch','ms_per_call','rtf','achieved_tflops_algorithmic')})"
