#!/usr/bin/env python3
"""Yardstick only (never on the product path): the vendor library's plain bf16 GEMM (torch.mm -> hipBLASLt / rocBLAS) on
the decoder's four projection shapes, next to this repository's engines with a PLAIN epilogue, same process, random data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib
lib = _lib.load()
DEV, BF, M = "cuda", torch.bfloat16, 12288
torch.manual_seed(0)

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2]

for name, N, K in (("qkv", 4096, 2048), ("o_proj", 2048, 2048), ("gate_up", 12288, 2048), ("down", 2048, 6144)):
    X = torch.randn(M, K, device=DEV).to(BF)
    W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    C = torch.empty(M, N, device=DEV, dtype=BF)
    Wt = W.t()
    t_vendor = timeit(lambda: torch.mm(X, Wt, out=C))
    s = torch.cuda.current_stream().cuda_stream
    t_ours = timeit(lambda: _lib.call("coda_gemm_bf16", X.data_ptr(), K, None, W.data_ptr(), K, C.data_ptr(), N, None, 0, M, N, K, 0, s))
    fl = 2.0 * M * N * K
    print(f"{name:8s} M={M} N={N} K={K}: vendor {t_vendor*1e3:7.1f} us {fl/t_vendor/1e9:7.1f} TF | this repo (plain epilogue) {t_ours*1e3:7.1f} us {fl/t_ours/1e9:7.1f} TF", flush=True)
