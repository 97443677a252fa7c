#!/usr/bin/env python3
"""In-kernel clock of the gate|up GEMM (MI355X_MICROARCH.md 'DVFS give-back' item 6): a diagnostic build of the
library (tools/probe/libcoda_hip_probe.so, -DCODA_CLOCK_PROBE) stamps s_memtime / s_memrealtime around every
workgroup's tile stream.  Runs >= 2 s of back-to-back launches on random data, then reads the last launch's stamps."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, LL, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
DEV, BF = "cuda", torch.bfloat16
M = 12288
SHAPES = {"gate_up": (12288, 2048, 2, 256), "down": (2048, 6144, 1, 192), "o_proj": (2048, 2048, 1, 192), "qkv_plain": (4096, 2048, 0, 256)}
torch.manual_seed(0)
tags = sys.argv[1].split(",") if len(sys.argv) > 1 else ["base"]
shapes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["gate_up"]
for shape in shapes:
    N, K, epi, bm = SHAPES[shape]
    X = torch.randn(M, K, device=DEV).to(BF)
    W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    cols = N // 2 if epi == 2 else N
    C = torch.randn(M, cols, device=DEV).to(BF)
    dbg = torch.zeros(2 * 256, device=DEV, dtype=torch.int64)
    s = torch.cuda.current_stream().cuda_stream
    for tag in tags:
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", f"libcoda_hip_probe_{tag}.so"))
        lib.coda_gemm_bf16_ex.argtypes = [P, LL, P, P, LL, P, LL, P, LL, I, I, I, I, I, I, P]
        lib.coda_gemm_bf16_ex.restype = I
        lib.coda_gemm_probe.argtypes = [P]
        variant = int(os.environ.get("GEMM_VARIANT", "0"))
        lib.coda_gemm_probe(dbg.data_ptr())

        def go():
            rc = lib.coda_gemm_bf16_ex(X.data_ptr(), K, None, W.data_ptr(), K, C.data_ptr(), cols, C.data_ptr() if epi == 1 else None,
                                       cols if epi == 1 else 0, M, N, K, epi, variant, 0, s)
            assert rc == 0, rc
        go()
        torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < 2.2:
            for _ in range(200):
                go()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            go()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        d = dbg.view(256, 2).cpu().double()
        clk = d[:, 0] / d[:, 1] * 100.0          # MHz
        cyc = d[:, 0]
        tiles_per_wg = (M // bm) * (N // 256) / 256
        mfma_cycles = tiles_per_wg * (K // 64) * (bm // 4) * 2 * 16   # per SIMD: tiles x K tiles x MFMAs per wave x 2 waves x 16 cycles
        print(f"{shape:9s} {tag:12s}: {us:7.1f} us/launch = {2.0 * M * N * K / us / 1e6:6.0f} TF | in-kernel clock median {clk.median():.0f} MHz "
              f"(min {clk.min():.0f}, max {clk.max():.0f}) | shader cycles per workgroup median {cyc.median():.0f} "
              f"-> MFMA pipe share {mfma_cycles / cyc.median():.3f}", flush=True)
