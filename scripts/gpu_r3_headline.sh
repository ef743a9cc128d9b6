#!/bin/bash
# headline kernel change: parity of the glow.py path first, then the default bench line and the bf16x3 / rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-h}
timeout 1500 python -m pytest tests/test_waveglow_gpu.py tests/test_full_size.py tests/test_conv1d_primitive.py tests/test_small_shape.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r3_${tag}_pytest_wg.log; tail -3 gpurun_out/r3_${tag}_pytest_wg.log
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-budget 5 --no-cpu-aggregate > gpurun_out/r3_${tag}_bench_f32.json 2> gpurun_out/r3_${tag}_bench_f32.err
python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r3_{sys.argv[1]}_bench_f32.json").read().strip().splitlines()[-1])
print("bench_f32", d["value"], d["ms_per_step"], d["roofline"])
PY
