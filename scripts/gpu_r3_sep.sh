#!/bin/bash
# fused separable WaveFlow layer with the DMA-staged, hand-scheduled fp32 loop: tests + the author's row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-sep}
timeout 1500 python -m pytest tests/test_waveflow.py tests/test_full_size.py tests/test_gemm_mode.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r3_${tag}_pytest.log; tail -4 gpurun_out/r3_${tag}_pytest.log
timeout 900 python scripts/bench_rows.py --rows waveflow_author --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err | cut -c1-330
