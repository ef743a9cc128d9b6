#!/bin/bash
# round 2, first GPU pass: full GPU suite, headline bench (fp32 + bf16), rocprof kernel stats of the fp32 line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2_a_pytest.log
python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r2_a_bench_f32.json 2> gpurun_out/r2_a_bench_f32.err
python bench.py --gpus 1 --steps 5 --warmup 2 --dtype bf16 --batch 32 --cpu-frames 0 > gpurun_out/r2_a_bench_bf16_b32.json 2> gpurun_out/r2_a_bench_bf16.err
tail -5 gpurun_out/r2_a_pytest.log; cat gpurun_out/r2_a_bench_f32.json; tail -3 gpurun_out/r2_a_bench_f32.err; cat gpurun_out/r2_a_bench_bf16_b32.json
