set -x; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_waveflow.py -m gpu -x -q -s 2>&1 | tail -8
python scripts/bench_rows.py --rows waveflow 2>&1 | tail -2
CTTS_WF_NO_FUSE=1 python scripts/bench_rows.py --rows waveflow 2>&1 | tail -1
python bench.py --dtype bf16 --steps 5 --warmup 2 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wf2 -o waveflow -- python $R/scripts/bench_rows.py --rows waveflow --steps 1 --warmup 0 > $R/gpurun_out/prof_wf2.log 2>&1
head -8 $R/gpurun_out/prof_wf2/waveflow_kernel_stats.csv | cut -c1-170
