mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_waveflow.py -m gpu -x -q -s 2>&1 | grep -E "rms rel|passed|failed|Error|assert" | tail -10
python scripts/bench_rows.py --rows waveflow_author 2>&1 | tail -1 | cut -c200-330
CTTS_WF_NO_FUSE=1 python scripts/bench_rows.py --rows waveflow_author 2>&1 | tail -1 | cut -c200-330
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wfa -o wfa -- python $R/scripts/bench_rows.py --rows waveflow_author --steps 1 --warmup 0 > $R/gpurun_out/prof_wfa.log 2>&1
python - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_wfa/wfa_kernel_stats.csv')))
for r in rows[:7]:
    print(r['Name'][:80].ljust(80), r['Calls'], f"{float(r['TotalDurationNs'])/1e6:8.2f} ms", f"{float(r['AverageNs'])/1e3:8.1f} us", r['Percentage'])
PY
