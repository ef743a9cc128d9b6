#!/bin/bash
# round 3, first pass on the ABI-4 tree: GPU suite, smoke, headline bench, rocprofv3 kernel stats, PMC passes of the headline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r3_a_pytest.log
tail -4 gpurun_out/r3_a_pytest.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/r3_a_smoke.log
python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r3_a_bench_f32.json 2> gpurun_out/r3_a_bench_f32.err
cut -c1-900 gpurun_out/r3_a_bench_f32.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_a_prof -o f32 -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 > gpurun_out/r3_a_prof.log 2>&1
ls gpurun_out/r3_a_prof | head
bash scripts/pmc3.sh f32 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing
bash scripts/pmc3.sh f32_bf16x3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing --gemm-mode bf16x3
bash scripts/pmc3.sh bf16_b32 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing --dtype bf16 --batch 32
ls -la gpurun_out/r3_pmc_*.json
