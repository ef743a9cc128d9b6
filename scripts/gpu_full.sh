set -x; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py > gpurun_out/bench_r1_final_f32.json 2> gpurun_out/bench_r1_final_f32.err; tail -2 gpurun_out/bench_r1_final_f32.err; cat gpurun_out/bench_r1_final_f32.json
python bench.py --dtype bf16 --cpu-frames 0 > gpurun_out/bench_r1_final_bf16.json 2>/dev/null; cat gpurun_out/bench_r1_final_bf16.json | cut -c1-400
python scripts/bench_rows.py > gpurun_out/rows_r1_final.jsonl 2>/dev/null; python scripts/bench_rows.py --rows waveflow_author >> gpurun_out/rows_r1_final.jsonl 2>/dev/null; cut -c1-330 gpurun_out/rows_r1_final.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f32_final -o f32 -- python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/prof_f32_final.log 2>&1
head -6 $R/gpurun_out/prof_f32_final/f32_kernel_stats.csv | cut -c1-160
