mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_waveglow_gpu.py -m gpu -x -q -k "bf16" 2>&1 | tail -4
for st in 3; do
CTTS_BF16_PP_STAGES=$st timeout 200 python bench.py --dtype bf16 --batch 8 --steps 5 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STAGES $st', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline'].get('res_skip_hbm',{}).get('mean_launch_ms'))"
done
