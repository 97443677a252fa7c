#!/usr/bin/env python3
"""A/B: N=2048 projections (o_proj, down_proj + residual) at M=12288 as ONE launch (auto engine) vs TWO launches:
rows [0, 8192) on the persistent 256x256 engine (exactly one round of 256 CUs) + the remaining rows on another engine.
Operands are rotated over several buffers so launches do not hit a warm cache."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib
lib = _lib.load()
DEV, BF = "cuda", torch.bfloat16
M, NB = 12288, 6

def run(X, W, C, R, m0, m1, variant):
    Xs, Cs, Rs = X[m0:m1], C[m0:m1], R[m0:m1]
    _lib.call("coda_gemm_bf16_ex", Xs.data_ptr(), X.stride(0), None, W.data_ptr(), W.stride(0), Cs.data_ptr(), C.stride(0),
              Rs.data_ptr(), R.stride(0), m1 - m0, W.shape[0], X.shape[1], 1, variant, 0, torch.cuda.current_stream().cuda_stream)

torch.manual_seed(0)
for name, N, K in (("o_proj", 2048, 2048), ("down", 2048, 6144)):
    Xs = [torch.randn(M, K, device=DEV).to(BF) for _ in range(NB)]
    Ws = [(torch.randn(N, K, device=DEV) * 0.02).to(BF) for _ in range(NB)]
    R = torch.randn(M, N, device=DEV).to(BF)
    C = torch.empty(M, N, device=DEV, dtype=BF)
    plans = {"one launch (auto)": [(0, M, 0)]}
    for split in (8192,):
        for v2 in (0, 1, 3, 4, 6):
            plans[f"[0,{split}) v3 + rest v{v2}"] = [(0, split, 3), (split, M, v2)]
    plans["[0,4096) v4 + rest v3"] = [(0, 4096, 4), (4096, M, 3)]
    # correctness of every plan against the single launch
    ref = None
    for pname, plan in plans.items():
        C.zero_()
        for m0, m1, v in plan:
            run(Xs[0], Ws[0], C, R, m0, m1, v)
        torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        else:
            assert torch.equal(ref, C) or (ref.float() - C.float()).abs().max() < 0.1, pname
    times = {p: [] for p in plans}
    for rnd in range(5):
        for pname, plan in plans.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(24):
                for m0, m1, v in plan:
                    run(Xs[it % NB], Ws[it % NB], C, R, m0, m1, v)
            e1.record()
            torch.cuda.synchronize()
            times[pname].append(e0.elapsed_time(e1) / 24 * 1e3)
    for pname in plans:
        t = sorted(times[pname])
        print(f"{name:7s} {pname:34s} median {t[len(t)//2]:7.1f} us  min {t[0]:7.1f} us", flush=True)
