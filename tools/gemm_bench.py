#!/usr/bin/env python3
"""A/B timing of the GEMM tile engines on the decoder's shapes (random data, interleaved rounds, one process)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
lib = _lib.load()
# A/B ACROSS BUILDS in one process: ALT_LIB=path/to/other/libcoda_hip.so; variant + 1000 runs on that library
ALT = {}
for n_, key_ in ((1, "ALT_LIB"), (2, "ALT_LIB2"), (3, "ALT_LIB3")):   # variant + 1000 / 2000 / 3000
    if os.environ.get(key_):
        import ctypes
        ALT[n_] = ctypes.CDLL(os.environ[key_])
        ALT[n_].coda_gemm_bf16_ex.restype, ALT[n_].coda_gemm_bf16_ex.argtypes = _lib.SIGNATURES["coda_gemm_bf16_ex"]


def run(X, W, C, R, epi, v=0):
    """v = engine number (coda_gemm_bf16_ex), + 100 = same engine with 8-byte epilogue stores."""
    args = (X.data_ptr(), X.stride(0), None, W.data_ptr(), W.stride(0), C.data_ptr(), C.stride(0),
            None if R is None else R.data_ptr(), 0 if R is None else R.stride(0), X.shape[0], W.shape[0], X.shape[1], epi,
            v % 100, _lib.GEMM_NARROW_STORES if (v % 1000) >= 100 else 0, torch.cuda.current_stream().cuda_stream)
    if v >= 1000:
        rc = ALT[v // 1000].coda_gemm_bf16_ex(*args)
        assert rc == 0, rc
    else:
        _lib.call("coda_gemm_bf16_ex", *args)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2"])]
    shapes = [("qkv", 4096, 2048, 0), ("o_proj", 2048, 2048, 1), ("gate_up", 12288, 2048, 2), ("down", 2048, 6144, 1)]
    if os.environ.get("GEMM_EXTRA"):
        shapes = [("plain12k", 12288, 2048, 0), ("plain12k_k8k", 12288, 8192, 0), ("swiglu_k8k", 12288, 8192, 2),
                  ("plain_n4k_k8k", 4096, 8192, 0)]
    torch.manual_seed(0)
    if os.environ.get("GEMM_ONLY"):
        shapes = [sh for sh in shapes if sh[0] in os.environ["GEMM_ONLY"].split(",")]
    for name, N, K, epi in shapes:
        X = torch.randn(M, K, device=DEV, dtype=torch.float32).to(BF)
        W = (torch.randn(N, K, device=DEV, dtype=torch.float32) * 0.02).to(BF)
        cols = N // 2 if epi == 2 else N
        R = torch.randn(M, N, device=DEV, dtype=torch.float32).to(BF) if epi == 1 else None
        outs = {}
        for v in variants:
            C = torch.zeros(M, cols, device=DEV, dtype=BF)
            run(X, W, C, R, epi, v)
            torch.cuda.synchronize()
            outs[v] = C.float()
        ref = X.float() @ W.float().t()
        if epi == 1:
            ref = ref.to(BF).float() + R.float()
        if epi == 2:
            from coda_amd.engine import interleave_gate_up  # noqa: F401
            g, u = ref.view(M, N // 32, 2, 16)[:, :, 0].reshape(M, -1), ref.view(M, N // 32, 2, 16)[:, :, 1].reshape(M, -1)
            ref = torch.nn.functional.silu(g.to(BF).float()).to(BF).float() * u.to(BF).float()
        msg = [f"{name:8s} M={M} N={N} K={K}"]
        for v in variants:
            err = (outs[v] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
            msg.append(f"v{v} relerr {err:.2e}")
        times = {v: [] for v in variants}
        for rnd in range(5):
            for v in variants:
                C = torch.empty(M, cols, device=DEV, dtype=BF)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                run(X, W, C, R, epi, v)
                e0.record()
                for _ in range(20):
                    run(X, W, C, R, epi, v)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 20)
        fl = 2.0 * M * N * K
        for v in variants:
            med = sorted(times[v])[len(times[v]) // 2]
            msg.append(f"v{v} {med*1e3:7.1f} us {fl/med/1e9:7.1f} TF (min {min(times[v])*1e3:.1f} us)")
        print(" | ".join(msg), flush=True)


if __name__ == "__main__":
    main()
