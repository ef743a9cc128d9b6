mkdir -p gpurun_out
python -m pytest tests/test_waveglow_gpu.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'])"
for v in 0 1; do   # 0 = m-block fastest mapping, 1 = XCD-pair mapping (default)
  if [ $v = 0 ]; then export CTTS_GEMM_NO_XCD_PAIR=1; else unset CTTS_GEMM_NO_XCD_PAIR; fi
  python bench.py --dtype bf16 --steps 5 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 PAIR=$v', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], (d['roofline'].get('res_hbm') or {}).get('mean_launch_ms'), (d['roofline'].get('skip_hbm') or {}).get('mean_launch_ms'))"
done
# FETCH_SIZE per in-layer launch under both mappings (own PMC pass each) -> profiles/r1_12_pmc_traffic.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 0 ]; then export CTTS_GEMM_NO_XCD_PAIR=1; else unset CTTS_GEMM_NO_XCD_PAIR; fi
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_p$v -o f -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_fetch_p$v.log 2>&1
done
