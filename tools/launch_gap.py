#!/usr/bin/env python3
"""Per-launch cost of a dependent chain of tiny kernels on one stream (what every extra launch in the step costs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import _lib
_lib.load()
dev = "cuda"
src = torch.zeros(8, 2048, device=dev, dtype=torch.bfloat16)
dst = torch.zeros(8, 2048, device=dev, dtype=torch.bfloat16)
rows = torch.arange(8, device=dev, dtype=torch.int32)
s = torch.cuda.current_stream().cuda_stream
def chain(n):
    for _ in range(n):
        _lib.call("coda_gather_rows", src.data_ptr(), 2048, rows.data_ptr(), None, 8, dst.data_ptr(), 2048, 2048, s)
chain(200); torch.cuda.synchronize()
for n in (1000, 4000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); chain(n); e1.record(); torch.cuda.synchronize()
    print(f"{n} dependent tiny launches: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch (ctypes + HIP launch included)")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(1000):
        _lib.call("coda_gather_rows", src.data_ptr(), 2048, rows.data_ptr(), None, 8, dst.data_ptr(), 2048, 2048, st)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"same chain replayed from a hipGraph: {e0.elapsed_time(e1) * 1e3 / 1000:.2f} us per kernel")
