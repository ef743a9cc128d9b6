#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/pmc3.sh tacotron scripts/bench_rows.py --rows tacotron --steps 1 --warmup 0
ls -la gpurun_out/r3_pmc_tacotron.json
