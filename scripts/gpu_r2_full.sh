#!/bin/bash
# round 2, full pass: GPU suite, smoke, headline bench lines (fp32 / bf16 B=32 / bf16x3), rocprofv3 kernel stats of the
# fp32 line, all secondary rows, persistent-decoder phase timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2_full_pytest.log
tail -4 gpurun_out/r2_full_pytest.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/r2_full_smoke.log
python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r2_full_bench_f32.json 2> gpurun_out/r2_full_bench_f32.err
python bench.py --gpus 1 --steps 5 --warmup 2 --dtype bf16 --batch 32 --cpu-frames 0 > gpurun_out/r2_full_bench_bf16_b32.json 2>/dev/null
python bench.py --gpus 1 --steps 5 --warmup 2 --dtype bf16x3 --cpu-frames 0 > gpurun_out/r2_full_bench_bf16x3.json 2>/dev/null
cut -c1-600 gpurun_out/r2_full_bench_f32.json; echo; cut -c1-300 gpurun_out/r2_full_bench_bf16_b32.json; echo; cut -c1-300 gpurun_out/r2_full_bench_bf16x3.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_full_prof -o f32 -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 > gpurun_out/r2_full_prof.log 2>&1
ls gpurun_out/r2_full_prof | head
timeout 900 python scripts/bench_rows.py --rows waveflow,waveflow_author,waveglow_ax,waveglow_ax_untts,tacotron,stft --steps 3 --warmup 1 2>gpurun_out/r2_full_rows.err > gpurun_out/r2_full_rows.jsonl
cut -c1-260 gpurun_out/r2_full_rows.jsonl
timeout 300 python scripts/profile_persistent.py > gpurun_out/r2_full_persistent_timeline.txt 2>&1; tail -8 gpurun_out/r2_full_persistent_timeline.txt
