#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes: runs each GEMM engine a few times on one shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

lib = _lib.load()
M, N, K, epi = 12288, int(os.environ.get("N", 12288)), int(os.environ.get("K", 2048)), int(os.environ.get("EPI", 2))
torch.manual_seed(0)
X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
R = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 1 else None
C = torch.empty(M, N // 2 if epi == 2 else N, device="cuda", dtype=torch.bfloat16)
for v in [int(a) for a in sys.argv[1:]] or [1, 2]:
    for _ in range(6):
        _lib.call("coda_gemm_bf16_ex", X.data_ptr(), K, None, W.data_ptr(), K, C.data_ptr(), C.stride(0),
                  None if R is None else R.data_ptr(), 0 if R is None else N, M, N, K, epi, v, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
