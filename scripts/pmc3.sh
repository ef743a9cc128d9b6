#!/bin/bash
# three PMC passes of one command (never with trace domains other than --kernel-trace; the program directly after --):
#   pmc3.sh <tag> <script.py> [args...]      -> gpurun_out/r6_pmc_<tag>_{fetch,write,sq}
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
prog=$1; shift
for pass in fetch write sq; do
  case $pass in
    fetch) ctr="FETCH_SIZE";;
    write) ctr="WRITE_SIZE";;
    sq) ctr="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
  esac
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/r6_pmc_${tag}_$pass -o p -- python3 $R/$prog "$@" > $R/gpurun_out/r6_pmc_${tag}_$pass.log 2>&1
  echo "pmc $tag $pass rc=$?"
done
python3 $R/scripts/pmc_summary.py --label "$prog $*" $R/gpurun_out/r6_pmc_${tag}_fetch $R/gpurun_out/r6_pmc_${tag}_write $R/gpurun_out/r6_pmc_${tag}_sq > $R/gpurun_out/r6_pmc_${tag}.json
# the raw csv files are large: keep only the summary
rm -rf $R/gpurun_out/r6_pmc_${tag}_fetch $R/gpurun_out/r6_pmc_${tag}_write $R/gpurun_out/r6_pmc_${tag}_sq
