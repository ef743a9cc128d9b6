#!/bin/bash
# The one GPU-box runner: gpurun -- 'bash scripts/gpu.sh <task> [args...]'.  Output under gpurun_out/<tag>_*.
#   tests [tag] [pytest args]   python -m pytest tests -m gpu <args>   (default: the whole suite)
#   py <tag> <script.py> [args] run a python script, log to gpurun_out/<tag>.log
#   bench <tag> [bench args]    python bench.py <args>, JSON line to gpurun_out/<tag>_bench.json
#   stats <tag> <prog> [args]   rocprofv3 --kernel-trace --stats of a python program -> gpurun_out/<tag>_kernel_stats.csv
#   pmc <tag> <prog> [args]     scripts/pmc3.sh: three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ wait/busy) + pmc_summary.py
#   ab <tag> <KNOB[=v]> [bench args]  bench.py with the environment knob off / on / off / on in one lease, one summary line each
#   micro <tag> <name>          scripts/micro/bin/<name> (cross-compiled in the build container) -> gpurun_out/<tag>.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
task=$1; shift
case "$task" in
  tests)
    tag=${1:-t}; shift
    args=("$@"); [ ${#args[@]} -eq 0 ] && args=(tests)
    timeout 3000 python -m pytest "${args[@]}" -m gpu -q -s 2>&1 | tail -400 > gpurun_out/${tag}_pytest.log
    tail -8 gpurun_out/${tag}_pytest.log ;;
  py)
    tag=$1; shift
    timeout 3000 python "$@" > gpurun_out/${tag}.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/${tag}.log ;;
  bench)
    tag=$1; shift
    timeout 1500 python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "rc=$?"
    tail -c 3000 gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err ;;
  stats)
    tag=$1; shift
    d=/tmp/prof_${tag}; rm -rf $d
    (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $d -o out --output-format csv -- python3 "$R/$1" "${@:2}" \
        > "$R/gpurun_out/${tag}_stats.log" 2>&1)
    f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/${tag}_kernel_stats.csv
    head -12 gpurun_out/${tag}_kernel_stats.csv; tail -3 gpurun_out/${tag}_stats.log ;;
  pmc)
    bash scripts/pmc3.sh "$@" ;;
  ab)
    tag=$1; knob=${2%%=*}; val=1; [[ "$2" == *=* ]] && val=${2#*=}; shift 2
    for rep in 1 2; do
      for mode in off on; do
        if [ $mode = on ]; then export $knob=$val; else unset $knob; fi
        timeout 900 python bench.py --no-rows --cpu-frames 0 "$@" > gpurun_out/${tag}_${mode}_$rep.json 2> gpurun_out/${tag}_${mode}_$rep.err
        python scripts/ab_summary.py gpurun_out/${tag}_${mode}_$rep.json "$knob=$val $mode $rep" | tee -a gpurun_out/${tag}_ab.txt
      done
    done ;;
  micro)
    tag=$1; shift
    timeout ${MICRO_TIMEOUT:-300} scripts/micro/bin/"$1" "${@:2}" > gpurun_out/${tag}.txt 2>&1; echo "rc=$?"; cat gpurun_out/${tag}.txt ;;
  *) echo "unknown task $task"; exit 2 ;;
esac
