#!/bin/bash
# round 3, end-of-round pass on the final tree: GPU suite, smoke, default bench.py (CPU baseline both legs, world-1 RCCL dry
# run), bf16 B=32 and split-bf16 lines, rocprofv3 kernel stats of the headline command, every secondary row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-full}
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_${tag}_pytest.log
tail -3 gpurun_out/r3_${tag}_pytest.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee gpurun_out/r3_${tag}_smoke.log
python bench.py > gpurun_out/r3_${tag}_bench_f32.json 2> gpurun_out/r3_${tag}_bench_f32.err
python bench.py --steps 5 --warmup 2 --dtype bf16 --batch 32 --cpu-frames 0 > gpurun_out/r3_${tag}_bench_bf16_b32.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --gemm-mode bf16x3 --cpu-frames 0 > gpurun_out/r3_${tag}_bench_f32_split_bf16x3.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --gemm-mode bf16x6 --cpu-frames 0 > gpurun_out/r3_${tag}_bench_f32_split_bf16x6.json 2>/dev/null
for f in bench_f32 bench_bf16_b32 bench_f32_split_bf16x3 bench_f32_split_bf16x6; do python - "$f" "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r3_{sys.argv[2]}_{sys.argv[1]}.json") if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"))
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_${tag}_prof -o f32 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-exchange-dry-run > $GRAFT_REPO_ROOT/gpurun_out/r3_${tag}_prof.log 2>&1
head -4 $GRAFT_REPO_ROOT/gpurun_out/r3_${tag}_prof/f32_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
timeout 1200 python scripts/bench_rows.py --rows waveflow,waveflow_author,waveglow_ax,waveglow_ax_untts,tacotron,stft --steps 3 --warmup 1 2>gpurun_out/r3_${tag}_rows.err > gpurun_out/r3_${tag}_rows.jsonl
cut -c1-150 gpurun_out/r3_${tag}_rows.jsonl
