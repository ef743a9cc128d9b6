mkdir -p gpurun_out
for d in 64 188 156; do
CTTS_BF16_DBG=$d timeout 200 python bench.py --dtype bf16 --batch 8 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DBG $d', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline'].get('res_skip_hbm',{}).get('mean_launch_ms'))"
done
