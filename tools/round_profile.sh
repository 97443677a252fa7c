#!/bin/bash
# Run on the GPU box from the repo root (gpurun -- 'bash tools/round_profile.sh'): GPU tests, the default bench line,
# and the three rocprofv3 passes that tools/summarize_prof.py reduces into profiles/ (kernel trace + stats;
# FETCH_SIZE and WRITE_SIZE in separate --pmc runs, as MI355X_MICROARCH.md prescribes).
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tests_gpu.log 2>&1 || { tail -20 gpurun_out/tests_gpu.log; exit 1; }
tail -2 gpurun_out/tests_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2>&1 || { tail -20 gpurun_out/bench_default.log; exit 1; }
tail -1 gpurun_out/bench_default.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-kernel-timer > $R/gpurun_out/prof/kt.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-kernel-timer --denoise-steps 8 > $R/gpurun_out/prof/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-kernel-timer --denoise-steps 8 > $R/gpurun_out/prof/pmc_write.log 2>&1 || exit 1
echo profiles collected
