mkdir -p gpurun_out
python -m pytest tests/test_waveglow_gpu.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'])"
for v in 0 1; do   # 0 = m-block fastest mapping, 1 = XCD-pair mapping (default)
  if [ $v = 0 ]; then export CTTS_GEMM_NO_XCD_PAIR=1; else unset CTTS_GEMM_NO_XCD_PAIR; fi
  python bench.py --dtype bf16 --steps 5 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 PAIR=$v', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline']['res_skip_hbm']['mean_launch_ms'])"
done
