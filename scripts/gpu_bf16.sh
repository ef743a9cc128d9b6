mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python bench.py --dtype bf16 --batch 8 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | tee gpurun_out/bench_r1_bf16_b8.json
python bench.py --dtype bf16 --batch 32 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | tee gpurun_out/bench_r1_bf16_b32.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bf16 -o bf16 -- python $R/bench.py --dtype bf16 --batch 8 --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/prof_bf16.log 2>&1
head -8 $R/gpurun_out/prof_bf16/bf16_kernel_stats.csv | cut -c1-140
