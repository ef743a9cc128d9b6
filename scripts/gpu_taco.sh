mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_tacotron.py tests/test_full_size.py -m gpu -x -q -k "tacotron or decoder or full_model" 2>&1 | tail -3
timeout 120 python scripts/bench_rows.py --rows tacotron 2>&1 | tail -1 | cut -c1-250
CTTS_TACO_NO_FUSE=1 timeout 120 python scripts/bench_rows.py --rows tacotron 2>&1 | tail -1 | cut -c1-200
