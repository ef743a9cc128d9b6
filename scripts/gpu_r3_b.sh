#!/bin/bash
# round 3, second pass: long-horizon Tacotron pins + the whole GPU suite, PMC passes of the secondary rows
# (config 4 WaveFlow dense, author's separable WaveFlow, ax notebook at B=1 and B=8), persistent-decoder timeline, rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tacotron_long.py tests/test_gemm_mode.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -150 > gpurun_out/r3_b_taco_long.log
tail -5 gpurun_out/r3_b_taco_long.log
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_b_pytest.log
tail -3 gpurun_out/r3_b_pytest.log
timeout 900 python scripts/bench_rows.py --rows waveflow,waveflow_author,waveglow_ax,waveglow_ax_untts,tacotron,stft --steps 3 --warmup 1 2>gpurun_out/r3_b_rows.err > gpurun_out/r3_b_rows.jsonl
cut -c1-200 gpurun_out/r3_b_rows.jsonl
timeout 300 python scripts/profile_persistent.py > gpurun_out/r3_b_persistent_timeline.txt 2>&1; tail -8 gpurun_out/r3_b_persistent_timeline.txt
bash scripts/pmc3.sh waveflow scripts/bench_rows.py --rows waveflow --steps 1 --warmup 0
bash scripts/pmc3.sh waveflow_author scripts/bench_rows.py --rows waveflow_author --steps 1 --warmup 0
bash scripts/pmc3.sh ax_b1 scripts/bench_rows.py --rows waveglow_ax --batches 1 --steps 1 --warmup 0
bash scripts/pmc3.sh ax_b8 scripts/bench_rows.py --rows waveglow_ax --batches 8 --steps 1 --warmup 0
ls -la gpurun_out/r3_pmc_*.json
