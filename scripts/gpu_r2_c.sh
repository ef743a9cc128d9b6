#!/bin/bash
# round 2: glow.py options (multispeaker / ReZero / grouped upsampling), speaker-dependent denoiser, ax notebook profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_waveglow_gpu.py tests/test_stft.py tests/test_waveglow_ax.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2_c_pytest.log
tail -15 gpurun_out/r2_c_pytest.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_c_prof -o ax -- python $R/scripts/bench_rows.py --rows waveglow_ax --steps 1 --warmup 1 > $R/gpurun_out/r2_c_prof.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r2_c_prof/**/*kernel_stats.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(r['Name'][:100].ljust(100), r['Calls'], f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
