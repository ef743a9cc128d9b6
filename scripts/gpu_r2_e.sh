#!/bin/bash
# round 2: persistent Tacotron decoder - parity (all decoder tests run through it by default), then the step-time row
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tacotron.py tests/test_full_size.py tests/test_stft.py -m gpu -x -q -k "tacotron or decoder or full_model or lockstep or stop or packed or denoiser" 2>&1 | tail -30 > gpurun_out/r2_e_pytest.log
tail -30 gpurun_out/r2_e_pytest.log
timeout 300 python scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 2>gpurun_out/r2_e_rows.err | cut -c1-700 | tee gpurun_out/r2_e_rows.jsonl
tail -3 gpurun_out/r2_e_rows.err
CTTS_TACO_NO_PERSIST=1 timeout 300 python scripts/bench_rows.py --rows tacotron --steps 3 --warmup 1 2>/dev/null | cut -c1-300
