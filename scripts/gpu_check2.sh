mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r1_04.json 2> gpurun_out/bench_r1_04.err; tail -2 gpurun_out/bench_r1_04.err; cat gpurun_out/bench_r1_04.json
python scripts/bench_rows.py --rows waveflow,stft 2>/dev/null | tee gpurun_out/rows_r1_b.jsonl
