#!/bin/bash
# round 2: ax 1-D with the cond interpolation inside the GATE epilogue - parity then the notebook row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_waveglow_ax.py tests/test_waveflow.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r2_i_pytest.log
timeout 600 python scripts/bench_rows.py --rows waveglow_ax --steps 3 --warmup 1 2>gpurun_out/r2_i_rows.err | tee gpurun_out/r2_i_rows.jsonl | cut -c1-420
python bench.py --steps 2 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fp32', round(d['ms_per_step'],2), r['mean_launch_ms'], r['frac'])"
