#!/bin/bash
# End-of-round pass: whole GPU suite (log headed by the HEAD it ran on), smoke, the driver's bench line (with rows), kernel stats
# of the headline.  gpurun -- 'bash scripts/gpu_end_of_round.sh <tag> <sha>'  (the sha is passed in: .git does not travel).
# Rule: no commit touching tests/ or csrc/ after this has run on the round's last HEAD.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-final}; sha=${2:-unknown}
bash scripts/gpu.sh tests ${tag}
sed -i "1i # HEAD ${sha} - scripts/gpu_end_of_round.sh ${tag}: python -m pytest tests -m gpu -q -s (whole GPU suite on MI355X)" gpurun_out/${tag}_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; tail -2 gpurun_out/${tag}_smoke.log
bash scripts/gpu.sh bench ${tag} | tail -c 600
bash scripts/gpu.sh stats ${tag}_f32 bench.py --steps 3 --warmup 1 --no-rows --cpu-frames 0 | head -6
