#!/usr/bin/env python3
"""In-kernel clock and cycles of the K loop of the few-rows tile engines (gemm_tile192.h) at the B = 1 shapes: a diagnostic build
(tools/build_probe.sh base) stamps s_memtime / s_memrealtime around the K loop of every workgroup.
usage: tools/build_probe.sh base "" && gpurun -- python tools/t192_probe.py"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, LL, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
DEV, BF = "cuda", torch.bfloat16
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libcoda_hip_probe_base.so"))
lib.coda_gemm_bf16_ex.argtypes = [P, LL, P, P, LL, P, LL, P, LL, I, I, I, I, I, I, P]
lib.coda_gemm_bf16_ex.restype = I
lib.coda_gemm_probe.argtypes = [P]
M = 384
torch.manual_seed(0)
for name, N, K, epi in (("gate_up", 12288, 2048, 2), ("o_proj", 2048, 2048, 0), ("down", 2048, 6144, 0)):
    X = torch.randn(M, K, device=DEV).to(BF)
    Ws = [(torch.randn(N, K, device=DEV) * 0.02).to(BF) for _ in range(12)]      # rotate: weights cold, as in the decoder
    cols = N // 2 if epi == 2 else N
    C = torch.zeros(M, cols, device=DEV, dtype=BF)
    dbg = torch.zeros(2 * 256, device=DEV, dtype=torch.int64)
    lib.coda_gemm_probe(dbg.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    for engine in (10, 11):
        def go(i):
            W = Ws[i % len(Ws)]
            rc = lib.coda_gemm_bf16_ex(X.data_ptr(), K, None, W.data_ptr(), K, C.data_ptr(), cols, None, 0, M, N, K, epi, engine, 0, s)
            assert rc == 0, rc
        t0 = time.time()
        i = 0
        while time.time() - t0 < 1.0:
            for _ in range(100):
                go(i); i += 1
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(120):
            go(i); i += 1
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 120 * 1e3
        d = dbg.view(256, 2).cpu().double()
        d = d[d[:, 1] > 0]
        clk = d[:, 0] / d[:, 1] * 100.0
        nk = K // 64
        print(f"{name:8s} engine {engine}: {us:6.1f} us/launch | K loop: {d[:, 1].median() / 100:6.2f} us = {d[:, 0].median():8.0f} shader cycles "
              f"({d[:, 0].median() / nk:6.0f} per K tile; MFMA issue alone 576) | in-kernel clock median {clk.median():.0f} MHz "
              f"(min {clk.min():.0f}, max {clk.max():.0f}) | workgroups stamped {len(d)}", flush=True)
