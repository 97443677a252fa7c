#!/bin/bash
# kernel trace of the B = 1 serving shape (one sequence, 128 + 256 tokens, window 32): gpurun -- 'bash tools/prof_b1.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt_b1 -o kt --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-kernel-timer --batch 1 > $R/gpurun_out/prof/kt_b1.log 2>&1 || exit 1
echo done
