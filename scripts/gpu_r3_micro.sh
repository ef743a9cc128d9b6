#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/micro/allgather_floor.hip -o /tmp/allgather_floor && timeout 200 /tmp/allgather_floor | grep "gather" > gpurun_out/r3_micro_allgather.txt
cat gpurun_out/r3_micro_allgather.txt
