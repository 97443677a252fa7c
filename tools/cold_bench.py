#!/usr/bin/env python3
"""Cache-state sensitivity of the projection GEMMs: the same launch with hot operands (one buffer set re-used) and
with the weights and/or activations rotated through more buffers than the 256 MB Infinity Cache holds - the state
the decoder's kernels actually see (28 layers of weights; activations written by the previous kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

lib = _lib.load()
DEV, BF = "cuda", torch.bfloat16
M = 12288
torch.manual_seed(0)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
for name, N, K, epi in (("qkv", 4096, 2048, 0), ("o_proj", 2048, 2048, 1), ("gate_up", 12288, 2048, 2), ("down", 2048, 6144, 1)):
    nW = max(2, int(400e6 / (N * K * 2)) + 1)
    nX = max(2, int(400e6 / (M * K * 2)) + 1)
    Ws = [(torch.randn(N, K, device=DEV) * 0.02).to(BF) for _ in range(nW)]
    Xs = [torch.randn(M, K, device=DEV).to(BF) for _ in range(nX)]
    cols = N // 2 if epi == 2 else N
    C = torch.randn(M, cols, device=DEV).to(BF)
    for v in variants:
        out = []
        for label, rw, rx in (("hot", 0, 0), ("W cold", 1, 0), ("X cold", 0, 1), ("both cold", 1, 1)):
            ts = []
            for rnd in range(3):
                n = 24
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for it in range(n + 2):
                    if it == 2:
                        e0.record()
                    X, W = Xs[it % nX if rx else 0], Ws[it % nW if rw else 0]
                    _lib.call("coda_gemm_bf16_ex", X.data_ptr(), K, None, W.data_ptr(), K, C.data_ptr(), cols,
                              C.data_ptr() if epi == 1 else None, cols if epi == 1 else 0, M, N, K, epi, v, 0,
                              torch.cuda.current_stream().cuda_stream)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / n * 1e3)
            out.append(f"{label}: {sorted(ts)[1]:.1f}")
        print(f"{name:8s} v{v} ({nW} W, {nX} X buffers) " + " | ".join(out) + " us", flush=True)
