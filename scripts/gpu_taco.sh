mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_tacotron.py tests/test_full_size.py -m gpu -x -q -k "tacotron or decoder or full_model or lockstep" 2>&1 | tail -3
timeout 300 python scripts/bench_rows.py --rows tacotron 2>&1 | tail -1 | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_taco9 -o tacotron -- python $R/scripts/bench_rows.py --rows tacotron --steps 1 --warmup 0 > $R/gpurun_out/prof_taco9.log 2>&1
python - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/prof_taco9/tacotron_kernel_stats.csv')))
for r in rows[:7]:
    print(r['Name'][:90].ljust(90), r['Calls'], f"{float(r['AverageNs'])/1e3:8.1f} us", r['Percentage'])
PY
