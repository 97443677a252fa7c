#!/usr/bin/env python3
"""A/B of the fused QKV + q/k-norm + RoPE launch: eight-wave engine (permuted weight rows) vs four-wave engine."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402
from coda_amd.engine import permute_qk_rows  # noqa: E402

_lib.load()
DEV, BF = "cuda", torch.bfloat16
n_q, n_kv, K, d = 16, 8, 2048, 128
N = (n_q + 2 * n_kv) * d
for B, L in ((32, 384), (64, 768), (8, 2048)):
    T = B * L
    torch.manual_seed(0)
    X = torch.randn(T, K, device=DEV).to(BF)
    W = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(BF)
    Wp = permute_qk_rows(W, n_q, n_kv)
    qw, kw = torch.ones(d, device=DEV, dtype=BF), torch.ones(d, device=DEV, dtype=BF)
    ang = torch.arange(L, device=DEV).float()[:, None] * (1.0 / (1e6 ** (torch.arange(0, d, 2, device=DEV).float() / d)))[None, :]
    emb = torch.cat((ang, ang), -1)
    cos, sin = emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous()
    outs = {}
    s = torch.cuda.current_stream().cuda_stream
    times = {}
    for name, w in (("coda_qkv_proj_rope", Wp), ("coda_qkv_proj_rope_w4", W)):
        out = torch.empty(T, N, device=DEV, dtype=BF)

        def go():
            _lib.call(name, X.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, qw.data_ptr(), kw.data_ptr(), cos.data_ptr(),
                      sin.data_ptr(), T, L, n_q, n_kv, K, 1e-6, s)
        go()
        torch.cuda.synchronize()
        outs[name] = out
        times[name] = (go, [])
    out2 = torch.empty(T, N, device=DEV, dtype=BF)

    def go_unfused():
        _lib.call("coda_gemm_bf16", X.data_ptr(), K, None, W.data_ptr(), K, out2.data_ptr(), N, None, 0, T, N, K, 0, s)
        _lib.call("coda_qknorm_rope", out2.data_ptr(), N, qw.data_ptr(), kw.data_ptr(), cos.data_ptr(), sin.data_ptr(), T, L,
                  n_q, n_kv, 1e-6, s)
    go_unfused()
    torch.cuda.synchronize()
    times["gemm + qknorm_rope"] = (go_unfused, [])
    for rnd in range(5):
        for name, (go, ts) in times.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            go()
            e0.record()
            for _ in range(20):
                go()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
    a, b = outs["coda_qkv_proj_rope"], outs["coda_qkv_proj_rope_w4"]
    same = (a == b).float().mean().item()
    fl = 2.0 * T * N * K
    print(f"T={T} L={L}: " + " | ".join(f"{n} {sorted(ts)[2]*1e3:7.1f} us {fl/sorted(ts)[2]/1e9:6.0f} TF" for n, (g, ts) in times.items())
          + f" | outputs equal {same:.5f}, unfused equal {(out2 == b).float().mean().item():.5f}", flush=True)
