#!/usr/bin/env python3
"""Fused lm-head + confidence at real vocabulary size: time per launch over candidate-row counts, optionally against a
second build of the library in the same process.  ROWS=65536,16384,... TEMP=0.0 ALT_LIB=path/to/libcoda_variant.so
(round 2: used to compare the shipped 128x128 kernel with the four-wave experiment, tools/experiments/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

if os.environ.get("ALT_LIB"):                 # a diagnostic build of the library (tools/alt/, not tracked)
    _lib.LIB_PATH = os.path.abspath(os.environ["ALT_LIB"])
DEV = "cuda"
V, H = 151936, 2048
rows = [int(x) for x in os.environ.get("ROWS", "65536,16384,4096,1024,256").split(",")]
temp = float(os.environ.get("TEMP", "0.0"))
torch.manual_seed(0)
T = max(rows)
hidden = torch.randn(T, H, device=DEV).to(torch.bfloat16)
W = (torch.randn(V, H, device=DEV) * 0.05).to(torch.bfloat16)
conf = torch.empty(T, device=DEV)
x0 = torch.empty(T, device=DEV, dtype=torch.int64)
ws = torch.empty(_lib.load().coda_lmhead_workspace_bytes(T, V), device=DEV, dtype=torch.uint8)
s = torch.cuda.current_stream().cuda_stream
for n in rows:
    src = torch.randperm(T, device=DEV)[:n].to(torch.int32)
    dst = torch.arange(n, device=DEV, dtype=torch.int32)
    cnt = torch.tensor([n], device=DEV, dtype=torch.int32)
    out = {}
    for name, cap in (("bound 4095", 4095), ("bound >= n", max(n, 4096))):
        def go():
            _lib.call("coda_lmhead_confidence", hidden.data_ptr(), H, src.data_ptr(), dst.data_ptr(), cnt.data_ptr(), cap,
                      W.data_ptr(), H, V, H, 0, temp, 0, 5, 0, conf.data_ptr(), x0.data_ptr(), ws.data_ptr(), s)
        go()
        torch.cuda.synchronize()
        reps = max(3, min(50, int(2e5 / n)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = (ms, x0[:n].clone(), conf[:n].clone())
        print(f"lm-head {n:6d} rows T={temp} {name:11s}: {ms:8.3f} ms  {2.0 * n * V * H / ms / 1e9:7.1f} TFLOP/s  "
              f"weight stream {V * H * 2 / ms / 1e6:6.0f} GB/s", flush=True)
    a, b = out["bound 4095"], out["bound >= n"]
    print(f"   tokens equal: {bool(torch.equal(a[1], b[1]))}   max |conf diff| {float((a[2] - b[2]).abs().max()):.2e}", flush=True)
