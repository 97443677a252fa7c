#!/usr/bin/env python3
"""Reduces rocprofv3 output (gpurun_out/prof/{kt,pmc_fetch,pmc_write}) to the summaries committed under profiles/.

  python tools/summarize_prof.py r01

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): separate --pmc passes for FETCH_SIZE and
WRITE_SIZE (KiB units); on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
read stream, so the read side is doubled; WRITE_SIZE is exact for 16-B/8-B streaming stores."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "")


kt = os.path.join(src, "kt", "kt_kernel_stats.csv")
rows = list(csv.DictReader(open(kt))) if os.path.exists(kt) else []
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, f"{tag}_kernel_stats.csv" if rows else os.devnull), "w") as f:
    f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
    for r in rows:
        if float(r["TotalDurationNs"]) / tot < 1e-4:
            continue
        f.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                f"{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f},{100*float(r['TotalDurationNs'])/tot:.2f}\n")
print(f"kernel stats: {len(rows)} kernels, {tot/1e6:.1f} ms total")


def counter_avg(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            key = short(r["Kernel_Name"]) + "|grid=" + r["Grid_Size"]
            acc[key][0] += float(r["Counter_Value"])
            acc[key][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


traffic = {}
fp = os.path.join(src, "pmc_fetch", "f_counter_collection.csv")
wp = os.path.join(src, "pmc_write", "w_counter_collection.csv")
if os.path.exists(fp) and os.path.exists(wp):
    fetch, write = counter_avg(fp, "FETCH_SIZE"), counter_avg(wp, "WRITE_SIZE")
    with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w") as f:
        f.write("kernel|grid,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,read_bytes_corrected(x2),write_bytes,hbm_bytes_per_launch\n")
        for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
            if not (k.startswith("gemm") or k.startswith("flash") or k.startswith("lmhead") or k.startswith("rmsnorm")
                    or k.startswith("qknorm") or k.startswith("confidence") or k.startswith("select") or k.startswith("build")
                    or k.startswith("embed")):
                continue
            fk, n = fetch[k]
            wk = write.get(k, (0.0, 0))[0]
            rb, wb = fk * 1024 * 2, wk * 1024
            traffic[k] = rb + wb
            f.write(f"{k},{n},{fk:.1f},{wk:.1f},{rb:.0f},{wb:.0f},{rb+wb:.0f}\n")
    # the gate|up projection (SwiGLU epilogue, template argument 2) on whichever tile engine ran it
    gate_up = [v for k, v in traffic.items() if k.startswith("gemm") and "_kernel<2" in k]
    out = {"source": f"profiles/{tag}_hbm_traffic.csv", "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB->bytes, per launch"}
    if gate_up:
        out["gate_up_hbm_bytes_per_launch"] = max(gate_up)
        out["gate_up_kernel"] = max((k for k in traffic if k.startswith("gemm") and "_kernel<2" in k), key=lambda k: traffic[k]).split("|")[0]
    json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print("traffic:", {k: f"{v/1e6:.1f} MB" for k, v in list(traffic.items())[:12]})


# ---- MFMA utilisation (tools/mfma_pmc.sh): per kernel, averaged over launches
m1 = os.path.join(src, "pmc_mfma1", "m_counter_collection.csv")
m2 = os.path.join(src, "pmc_mfma2", "m_counter_collection.csv")
if os.path.exists(m1):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for path in (m1, m2):
        if not os.path.exists(path):
            continue
        seen = set()
        with open(path) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"]) + "|grid=" + r["Grid_Size"]
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                did = (path, r.get("Dispatch_Id"))
                if did not in seen and path == m1:
                    seen.add(did)
                    dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    dur[k][1] += 1
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
             "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"]
    util = {}
    with open(os.path.join(dst, f"{tag}_mfma_pmc.csv"), "w") as f:
        f.write("kernel|grid,launches,avg_us_under_pmc," + ",".join(names) + ",mfma_util(=MFMA_BUSY/(GUI_ACTIVE/8 XCDs*1024 SIMDs)),lds_conflict_share(=BANK_CONFLICT/IDX_ACTIVE)\n")
        for k in sorted(acc, key=lambda k: -dur[k][0]):
            if not any(k.startswith(p) for p in ("gemm", "flash", "lmhead", "rmsnorm", "confidence")):
                continue
            v = {n: (acc[k][n][0] / acc[k][n][1] if acc[k][n][1] else float("nan")) for n in names}
            u = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128)   # GUI_ACTIVE is summed over the 8 XCDs; MFMA_BUSY over the 1024 SIMDs if v["GRBM_GUI_ACTIVE"] == v["GRBM_GUI_ACTIVE"] and v["GRBM_GUI_ACTIVE"] else float("nan")
            c = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"] if v["SQ_LDS_IDX_ACTIVE"] == v["SQ_LDS_IDX_ACTIVE"] and v["SQ_LDS_IDX_ACTIVE"] else float("nan")
            util[k] = u
            f.write(f'"{k}",{dur[k][1]},{dur[k][0]/max(dur[k][1],1):.1f},' + ",".join(f"{v[n]:.0f}" for n in names) + f",{u:.4f},{c:.4f}\n")
    tj = os.path.join(dst, "traffic.json")
    out = json.load(open(tj)) if os.path.exists(tj) else {}
    gu = [k for k in util if k.startswith("gemm") and "_kernel<2" in k and util[k] == util[k]]
    if gu:
        best = max(gu, key=lambda k: dur[k][0])
        out["gate_up_mfma_util"] = util[best]
        out["gate_up_mfma_util_source"] = f"profiles/{tag}_mfma_pmc.csv ({best})"
        json.dump(out, open(tj, "w"), indent=1)
    print("mfma util:", {k.split("|")[0][:40]: round(u, 3) for k, u in list(util.items())[:10]})
