#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gemm_mode.py -m gpu -q 2>&1 | tail -2
python bench.py --gemm-mode bf16x3 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fp32-layout split: ms/step', round(d['ms_per_step'],2), 'in-layer', r['mean_launch_ms'])"
CTTS_F32_GEMM_MODE=bf16x3 timeout 900 python scripts/bench_rows.py --rows waveflow,waveglow_ax --steps 3 --warmup 1 2>/dev/null | cut -c1-260
