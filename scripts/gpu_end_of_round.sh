#!/bin/bash
# end-of-round pass (tag r4_25: the last one of round 4): whole GPU suite, smoke, the driver's bench line (with rows), kernel stats of the headline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash scripts/gpu.sh tests r4_25
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_25_smoke.log 2>&1; tail -2 gpurun_out/r4_25_smoke.log
bash scripts/gpu.sh bench r4_25 | tail -c 600
bash scripts/gpu.sh stats r4_25_f32 bench.py --steps 3 --warmup 1 --no-rows --cpu-frames 0 | head -6
