#!/usr/bin/env python3
"""Times coda_confidence_from_logits (materialised-logits path: top-p / top-k / hooks) at vocabulary size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib  # noqa: E402

if os.environ.get("ALT_LIB"):                 # another build of the library (tools/alt/, not tracked): A/B across builds
    _lib.LIB_PATH = os.path.abspath(os.environ["ALT_LIB"])
_lib.load()
N, V = int(os.environ.get("ROWS", "8192")), 151936
torch.manual_seed(0)
scale = float(os.environ.get("SCALE", "3.0"))
logits = (torch.randn(N, V, device="cuda") * scale).to(torch.bfloat16)
conf = torch.zeros(N, device="cuda")
x0 = torch.zeros(N, dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
gb = N * V * 2 / 1e9
for name, alg, T, top_p, top_k in (("greedy entropy", 2, 0.0, 0.0, 0), ("greedy maskgit", 0, 0.0, 0.0, 0), ("top_p 0.9 greedy", 2, 0.0, 0.9, 0),
                                   ("top_p 0.9 T=0.1", 2, 0.1, 0.9, 0), ("T=0.7 no filter", 0, 0.7, 0.0, 0), ("top_k 50 greedy", 2, 0.0, 0.0, 50),
                                   ("top_p 0.9 T=0.7", 0, 0.7, 0.9, 0)):
    def go():
        _lib.call("coda_confidence_from_logits", logits.data_ptr(), V, 1, N, V, None, None, alg, T, top_p, top_k, None, 1234, 0,
                  conf.data_ptr(), x0.data_ptr(), s)
    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name:18s}: {ms:8.2f} ms for {N} rows = {gb / ms:6.2f} TB/s of logits (one pass = {gb:.2f} GB)", flush=True)
