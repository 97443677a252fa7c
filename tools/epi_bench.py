#!/usr/bin/env python3
"""What an epilogue costs: the same GEMM shape with the PLAIN / RESIDUAL / F32 epilogues, interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

lib = _lib.load()
DEV, BF = "cuda", torch.bfloat16
M = 12288
torch.manual_seed(0)
for name, N, K in (("o_proj", 2048, 2048), ("down", 2048, 6144)):
    X = torch.randn(M, K, device=DEV).to(BF)
    W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    R = torch.randn(M, N, device=DEV).to(BF)
    C = torch.empty(M, N, device=DEV, dtype=BF)
    res = {}
    for rnd in range(5):
        for epi, label in ((0, "plain"), (1, "residual"), (11, "residual in place")):
            def go():
                r = None if epi == 0 else (C if epi == 11 else R)
                _lib.call("coda_gemm_bf16", X.data_ptr(), K, None, W.data_ptr(), K, C.data_ptr(), N, None if r is None else r.data_ptr(),
                          0 if r is None else N, M, N, K, min(epi, 1), torch.cuda.current_stream().cuda_stream)
            go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                go()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(label, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(name, " | ".join(f"{k}: {sorted(v)[2]:.1f} us" for k, v in res.items()), flush=True)
