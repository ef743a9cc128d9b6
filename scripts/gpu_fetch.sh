# FETCH_SIZE / WRITE_SIZE of the headline command, one counter per pass -> profiles/*pmc_traffic.json
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_g -o f -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_fetch_g.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_g -o w -- python $R/bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-kernel-timing > $R/gpurun_out/pmc_write_g.log 2>&1
ls $R/gpurun_out/pmc_fetch_g $R/gpurun_out/pmc_write_g
