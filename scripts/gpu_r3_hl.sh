#!/bin/bash
# headline launch alone, product loop and bounding experiments (scripts/micro/headline_gemm.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for b in scripts/micro/bin/hl_*; do
  n=$(basename $b); timeout 120 $b ${n#hl_}
done > gpurun_out/r3_headline_gemm_experiments.txt 2>&1
cat gpurun_out/r3_headline_gemm_experiments.txt
