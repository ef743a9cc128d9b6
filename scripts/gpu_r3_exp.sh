#!/bin/bash
# upper-bound experiments on the small conv-GEMM main loop (binaries built from temporarily patched sources)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for b in scripts/micro/bin/exp_*; do
  echo "== $b"
  timeout 120 $b 2>&1 | grep -v "^rep [01]"
done > gpurun_out/r3_small_gemm_experiments.txt 2>&1
cat gpurun_out/r3_small_gemm_experiments.txt | grep "==\|rep\|main loop\|blocks,"
