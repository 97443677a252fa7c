#!/usr/bin/env python3
"""In-kernel clock and cycles of the attention kernel (diagnostic build tools/probe/libcoda_hip_probe_<tag>.so)."""
import ctypes
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1].split(",") if len(sys.argv) > 1 else ["ls1"]
P, LL, I, F = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_float
n_q, n_kv, d = 16, 8, 128
for B, L in ((32, 384), (8, 2048)):
    T = B * L
    torch.manual_seed(0)
    qkv = torch.randn(T, (n_q + 2 * n_kv) * d, device="cuda").to(torch.bfloat16)
    out = torch.empty(T, n_q * d, device="cuda", dtype=torch.bfloat16)
    dbg = torch.zeros(2 * 65536, device="cuda", dtype=torch.int64)
    for tag in tags:
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", f"libcoda_hip_probe_{tag}.so"))
        lib.coda_flash_attn_noncausal.argtypes = [P, LL, P, LL, I, I, I, I, F, P]
        lib.coda_attn_probe.argtypes = [P]
        lib.coda_attn_probe(dbg.data_ptr())
        s = torch.cuda.current_stream().cuda_stream

        def go():
            rc = lib.coda_flash_attn_noncausal(qkv.data_ptr(), qkv.stride(0), out.data_ptr(), n_q * d, B, L, n_q, n_kv, 1.0 / math.sqrt(d), s)
            assert rc == 0
        go()
        torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < 2.2:
            for _ in range(500):
                go()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            go()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        dd = dbg.view(-1, 2).cpu().double()
        dd = dd[dd[:, 1] > 0]
        clk = dd[:, 0] / dd[:, 1] * 100.0
        ntile = (L + 63) // 64
        nqb = (L + 127) // 128
        wgs = dd.shape[0]
        blocks_per_wg = B * n_q * nqb / wgs
        mfma = blocks_per_wg * ntile * 32 * 32          # cycles of this wave's own MFMAs
        print(f"attn B={B} L={L} {tag}: {us:7.1f} us = {4.0 * L * L * d * n_q * B / us / 1e6:6.0f} TF | {wgs} workgroups, clock median {clk.median():.0f} MHz | "
              f"cycles per workgroup median {dd[:, 0].median():.0f} (max {dd[:, 0].max():.0f}); own MFMA cycles per wave {mfma:.0f} "
              f"= {mfma / dd[:, 0].median():.3f} of the lifetime (x2 waves per SIMD)", flush=True)
