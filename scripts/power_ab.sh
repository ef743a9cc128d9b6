#!/bin/bash
# Energy A/B of one environment knob ("measure energy, not only time"): bench.py with the knob off / on / off / on
# while `rocm-smi --showpower` is sampled beside it; per run: ms/step, mean board power over the busy samples, joules per step.
#   gpurun -- 'bash scripts/power_ab.sh <tag> <KNOB[=v]> [bench args]'   -> gpurun_out/<tag>_power_ab.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; knob=${2%%=*}; val=1; [[ "$2" == *=* ]] && val=${2#*=}; shift 2
out=gpurun_out/${tag}_power_ab.txt
echo "# bench.py --no-rows --cpu-frames 0 $* ; knob $knob=$val ; power = mean of the rocm-smi --showpower samples >= 70 % of the run's maximum" > $out
for rep in 1 2; do
  for mode in off on; do
    if [ $mode = on ]; then export $knob=$val; else unset $knob; fi
    pw=gpurun_out/${tag}_${mode}_${rep}.power
    ( while true; do rocm-smi --showpower --csv 2>/dev/null | grep -i "^card" | head -1; sleep 0.15; done ) > $pw &
    SP=$!
    timeout 900 python bench.py --no-rows --cpu-frames 0 "$@" > gpurun_out/${tag}_${mode}_${rep}.json 2> gpurun_out/${tag}_${mode}_${rep}.err
    kill $SP 2>/dev/null; wait $SP 2>/dev/null
    python3 - "$pw" gpurun_out/${tag}_${mode}_${rep}.json "$knob=$val $mode $rep" >> $out <<'PY'
import json, sys
vals = []
for line in open(sys.argv[1]):
    for tok in line.strip().split(",")[1:]:
        try:
            vals.append(float(tok)); break
        except ValueError:
            pass
d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith('{"metric"')][-1])   # (RCCL prints its banner behind the line)
busy = [v for v in vals if v >= 0.7 * max(vals)] if vals else []
p = sum(busy) / len(busy) if busy else float("nan")
ms = d["ms_per_step"]
r = d.get("roofline", {})
print(f"{sys.argv[3]:28s} {ms:8.2f} ms/step  in-layer {r.get('mean_launch_ms', float('nan')):.4f} ms  power {p:6.1f} W over {len(busy)}/{len(vals)} samples  "
      f"{p * ms / 1e3:7.2f} J/step")
PY
  done
done
cat $out
