set -x; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -2
python bench.py > gpurun_out/bench_r1_final_f32.json 2> gpurun_out/bench_r1_final_f32.err; tail -2 gpurun_out/bench_r1_final_f32.err; cat gpurun_out/bench_r1_final_f32.json
python bench.py --dtype bf16 --cpu-frames 0 > gpurun_out/bench_r1_final_bf16.json 2>/dev/null; cat gpurun_out/bench_r1_final_bf16.json
