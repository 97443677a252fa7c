#!/usr/bin/env python3
"""Greedy confidence of n candidate rows at real vocabulary size: the fused lm-head kernel (logits never leave the
accumulators) against the materialised route (row gather, counted GEMM on the four-wave engine writing bf16 logits,
confidence kernel over them).  ROWS=65536,16384,..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

DEV = "cuda"
V, H = 151936, 2048
rows = [int(x) for x in os.environ.get("ROWS", "65536,16384,8192,4096,1024").split(",")]
alg = int(os.environ.get("ALG", "1"))
torch.manual_seed(0)
T = max(rows)
hidden = torch.randn(T, H, device=DEV).to(torch.bfloat16)
W = (torch.randn(V, H, device=DEV) * 0.05).to(torch.bfloat16)
ws = torch.empty(_lib.load().coda_lmhead_workspace_bytes(T, V), device=DEV, dtype=torch.uint8)
s = torch.cuda.current_stream().cuda_stream


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for n in rows:
    src = torch.randperm(T, device=DEV)[:n].to(torch.int32)
    dst = torch.arange(n, device=DEV, dtype=torch.int32)
    cnt = torch.tensor([n], device=DEV, dtype=torch.int32)
    conf_f, x0_f = torch.empty(T, device=DEV), torch.empty(T, device=DEV, dtype=torch.int64)
    conf_u, x0_u = torch.empty(T, device=DEV), torch.empty(T, device=DEV, dtype=torch.int64)
    hc = torch.empty(n, H, device=DEV, dtype=torch.bfloat16)
    logits = torch.empty(n, V, device=DEV, dtype=torch.bfloat16)

    def fused():
        _lib.call("coda_lmhead_confidence", hidden.data_ptr(), H, src.data_ptr(), dst.data_ptr(), cnt.data_ptr(), n,
                  W.data_ptr(), H, V, H, alg, 0.0, 0, 5, 0, conf_f.data_ptr(), x0_f.data_ptr(), ws.data_ptr(), s)

    def unfused():
        _lib.call("coda_gather_rows", hidden.data_ptr(), H, src.data_ptr(), cnt.data_ptr(), n, hc.data_ptr(), H, H, s)
        _lib.call("coda_gemm_bf16_counted", hc.data_ptr(), H, cnt.data_ptr(), W.data_ptr(), H, logits.data_ptr(), V, n, V, H, s)
        _lib.call("coda_confidence_from_logits", logits.data_ptr(), V, 1, n, V, cnt.data_ptr(), dst.data_ptr(), alg, 0.0, 0.0, 0,
                  0, 5, 0, conf_u.data_ptr(), x0_u.data_ptr(), s)
    reps = max(3, min(50, int(2e5 / n)))
    tf, tu = timed(fused, reps), timed(unfused, reps)
    same = bool(torch.equal(x0_f[:n], x0_u[:n]))
    print(f"{n:6d} rows: fused {tf:8.3f} ms ({2.0 * n * V * H / tf / 1e9:6.1f} TFLOP/s) | gather + GEMM + confidence {tu:8.3f} ms "
          f"({n * V * 2 / 1e9:5.1f} GB of logits) | tokens equal {same}, max |conf diff| {float((conf_f[:n] - conf_u[:n]).abs().max()):.2e}", flush=True)
    del logits, hc
    torch.cuda.empty_cache()
