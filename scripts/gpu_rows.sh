set -x; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python scripts/bench_rows.py > gpurun_out/rows_r1.jsonl 2> gpurun_out/rows_r1.err; tail -3 gpurun_out/rows_r1.err; cat gpurun_out/rows_r1.jsonl
cd /tmp && export TMPDIR=/tmp
for row in waveflow tacotron stft; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$row -o $row -- python $R/scripts/bench_rows.py --rows $row --steps 1 --warmup 0 > $R/gpurun_out/prof_$row.log 2>&1
  head -8 $R/gpurun_out/prof_$row/${row}_kernel_stats.csv | cut -c1-160
done
