#!/bin/bash
# The three rocprofv3 passes behind profiles/<tag>_kernel_stats.csv and <tag>_hbm_traffic.csv (run on the GPU box from the
# repo root): kernel trace + stats over one full C2 generate; FETCH_SIZE and WRITE_SIZE in separate --pmc runs.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-kernel-timer > $R/gpurun_out/prof/kt.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-kernel-timer --denoise-steps 8 > $R/gpurun_out/prof/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-kernel-timer --denoise-steps 8 > $R/gpurun_out/prof/pmc_write.log 2>&1 || exit 1
echo profiles collected
