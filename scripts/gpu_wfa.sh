mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_waveflow.py -m gpu -x -q 2>&1 | tail -3
python scripts/bench_rows.py --rows waveflow_author 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wfa -o wfa -- python $R/scripts/bench_rows.py --rows waveflow_author --steps 1 --warmup 0 > $R/gpurun_out/prof_wfa.log 2>&1
head -12 $R/gpurun_out/prof_wfa/wfa_kernel_stats.csv | cut -d, -f2-5 
