#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CTTS_F32_GEMM_MODE=bf16x3 timeout 900 python -m pytest tests/test_waveflow.py tests/test_full_size.py -m gpu -q -s 2>&1 | grep -E "^FAILED|passed|failed|rms rel" | tail -14
CTTS_F32_GEMM_MODE=bf16x3 timeout 900 python scripts/bench_rows.py --rows waveflow,waveflow_author,waveglow_ax --steps 3 --warmup 1 2>/dev/null | tee gpurun_out/r2_m_rows.jsonl | cut -c1-260
python bench.py --gemm-mode bf16x3 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null > gpurun_out/r2_m_bench_f32_split.json; cut -c1-330 gpurun_out/r2_m_bench_f32_split.json
