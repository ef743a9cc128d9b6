#!/bin/bash
# default bench.py (CPU baseline with the concurrent-instances leg before the GPU is touched, world-1 RCCL dry run),
# bf16-vs-reference numbers, rocprofv3 kernel stats of the config-5 row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python bench.py ) > gpurun_out/r3_j_bench_default.json 2> gpurun_out/r3_j_bench_default.err
tail -4 gpurun_out/r3_j_bench_default.err; cut -c1-1500 gpurun_out/r3_j_bench_default.json; echo
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_j_bench_default.json") if l.startswith("{")][-1])
print("cpu_baseline:", json.dumps(d["cpu_baseline"])[:1500])
print("exchange:", json.dumps(d.get("exchange"))[:600])
PY
timeout 1200 python -m pytest tests/test_full_size.py tests/test_waveglow_gpu.py tests/test_gemm_mode.py tests/test_conv1d_primitive.py -m gpu -q -s 2>&1 | grep -i "bf16\|passed\|failed\|error\|rms rel" | tail -30
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_j_prof_taco -o taco -- python3 $GRAFT_REPO_ROOT/scripts/bench_rows.py --rows tacotron --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_j_prof_taco.log 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/r3_j_prof_taco/taco_kernel_stats.csv | cut -c1-200
