#!/bin/bash
# rocprofv3 counter passes for MFMA utilisation over one C2 generate (run on the GPU box from the repo root:
# gpurun -- 'bash tools/mfma_pmc.sh').  Counters only with --kernel-trace; the program sits directly after `--`.
# Reduced by tools/summarize_prof.py <tag> into profiles/<tag>_mfma_pmc.csv.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --cpu-steps 0 --no-kernel-timer --denoise-steps 16"
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/prof/pmc_mfma1 -o m --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof/pmc_mfma1.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/prof/pmc_mfma2 -o m --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof/pmc_mfma2.log 2>&1 || exit 1
echo mfma counters collected
