#!/bin/bash
# rocprofv3 SQ counter passes over tools/attn_bench.py (run on the GPU box from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_a1 -o p --output-format csv -- python3 $R/tools/attn_bench.py > $R/gpurun_out/pmc_a1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_SALU --kernel-trace -d $R/gpurun_out/pmc_a2 -o p --output-format csv -- python3 $R/tools/attn_bench.py > $R/gpurun_out/pmc_a2.log 2>&1
cd $R
python3 - <<PY
import csv,collections
for d in ("pmc_a1","pmc_a2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"gpurun_out/{d}/p_counter_collection.csv")):
        if "flash" not in r["Kernel_Name"]: continue
        acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g,cs in acc.items():
        print(d,"grid",g,{c:round(sum(v)/len(v)/1e6,2) for c,v in cs.items()})
PY
