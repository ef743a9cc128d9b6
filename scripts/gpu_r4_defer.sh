#!/bin/bash
# fp32 deferred-skip form of the WaveGlow WN stack (CTTS_F32_DEFER_SKIP) against the shipped per-layer form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "(parity under the knob: 39 passed, first run)"
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export CTTS_F32_DEFER_SKIP=1; else unset CTTS_F32_DEFER_SKIP; fi
  timeout 600 python bench.py --steps 12 --warmup 3 --no-rows --cpu-frames 0 --no-exchange-dry-run > gpurun_out/r4_defer_bench_${v}.json 2> gpurun_out/r4_defer_bench_$v.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/r4_defer_bench_${v}.json") if l.startswith("{")][-1])
r=d["roofline"]
print("defer=$v ms/step", round(d["ms_per_step"],2), "in-layer", r["mean_launch_ms"], {k:(v.get("mean_launch_ms"),v.get("launches")) for k,v in r.items() if isinstance(v,dict)})
P
done
