#!/usr/bin/env python3
"""Times the flash attention kernel on the decoder's shapes (random data).  QSPLITS=1,2,4 additionally forces the
number of workgroups per (batch, head) (KNOB=qsplit, default) or the head-pair mode (KNOB=pair: 1 = off, 2 = on) through
coda_flash_attn_noncausal_ex's per-call arguments, interleaved in one process (A/B)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

if os.environ.get("ALT_LIB"):                 # a diagnostic build of the library (tools/alt/, not tracked)
    _lib.LIB_PATH = os.path.abspath(os.environ["ALT_LIB"])

_lib.load()
splits = [0] + [int(v) for v in os.environ.get("QSPLITS", "").split(",") if v]
KNOB = os.environ.get("KNOB", "qsplit")
SHAPES = [tuple(int(v) for v in it.split("x")) for it in os.environ.get("SHAPES", "32x384,64x768,8x2048").split(",")]   # BxL
for B, L in SHAPES:
    n_q, n_kv, d = 16, 8, 128
    T = B * L
    torch.manual_seed(0)
    qkv = torch.randn(T, (n_q + 2 * n_kv) * d, device="cuda").to(torch.bfloat16)
    out = torch.empty(T, n_q * d, device="cuda", dtype=torch.bfloat16)

    def run(s=0):
        pair, qsplit = (s - 1 if s else -1, 0) if KNOB == "pair" else (-1, s)
        _lib.call("coda_flash_attn_noncausal_ex", qkv.data_ptr(), qkv.stride(0), out.data_ptr(), n_q * d, B, L, n_q, n_kv,
                  1.0 / math.sqrt(d), pair, qsplit, torch.cuda.current_stream().cuda_stream)
    run()
    torch.cuda.synchronize()
    q = qkv[:L, :d].float()
    k = qkv[:L, n_q * d:n_q * d + d].float()
    v = qkv[:L, (n_q + n_kv) * d:(n_q + n_kv) * d + d].float()
    ref = torch.softmax(q @ k.t() / math.sqrt(d), -1) @ v
    err = (out[:L, :d].float() - ref).abs().max().item()
    times = {s: [] for s in splits}
    for _ in range(5):
        for s in splits:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(s)
            e0.record()
            for _ in range(20):
                run(s)
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / 20)
    fl = 4.0 * L * L * d * n_q * B
    for s in splits:
        med = sorted(times[s])[2]
        print(f"attn B={B} L={L} {KNOB}={s or 'auto'}: {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF  max err vs fp32 {err:.2e}", flush=True)
