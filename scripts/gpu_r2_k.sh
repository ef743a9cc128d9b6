#!/bin/bash
# round 2: split-bf16 main loop of the fp32 conv-GEMM - parity of every fp32 path under it, then the rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CTTS_F32_GEMM_MODE=bf16x3 timeout 1500 python -m pytest tests -m gpu -x -q -k "not staging_variants and not block_shapes" 2>&1 | tail -15 | tee gpurun_out/r2_k_pytest.log
python bench.py --gemm-mode bf16x3 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fp32-layout split: ms/step', round(d['ms_per_step'],2), 'in-layer', r['mean_launch_ms'])"
CTTS_F32_GEMM_MODE=bf16x3 timeout 900 python scripts/bench_rows.py --rows waveflow,waveflow_author,waveglow_ax,stft --steps 3 --warmup 1 2>gpurun_out/r2_k_rows.err | tee gpurun_out/r2_k_rows.jsonl | cut -c1-330
