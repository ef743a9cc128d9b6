#!/bin/bash
# headline A/B of an experimental knob (env name in $1), interleaved twice in one lease
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
knob=$1
for rep in 1 2; do
  for mode in off on; do
    if [ $mode = on ]; then export $knob=${2:-1}; else unset $knob; fi
    timeout 600 python bench.py --steps 10 --warmup 3 --no-rows --cpu-frames 0 > gpurun_out/r4_13_${mode}_$rep.json 2> gpurun_out/r4_13_${mode}_$rep.err
    python - <<P
import json
for l in open("gpurun_out/r4_13_${mode}_$rep.json"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("$knob $mode $rep", round(d["ms_per_step"],2), "in", r["mean_launch_ms"], r["frac"], "res", r["res_skip_mfma"]["mean_launch_ms"], r["res_skip_mfma"]["frac"], "skip", r["skip_mfma"]["mean_launch_ms"], r["skip_mfma"]["frac"], "step", r["step_mfma_algorithmic"]["frac"])
P
  done
done
