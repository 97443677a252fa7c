#!/usr/bin/env python3
"""Experiment: one forward of the full batch on one stream vs the two half batches on two streams (kernel tails,
ramps and half-empty tile rounds of one half overlap with the other half's kernels).  C2 shape, 28 layers."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import CoDAConfig, CoDALanguageModel  # noqa: E402
from coda_amd.engine import HipEngine  # noqa: E402

dev = torch.device("cuda", 0)
cfg = CoDAConfig(num_hidden_layers=int(os.environ.get("LAYERS", "28")))
torch.manual_seed(1234)
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    model = CoDALanguageModel(cfg)
torch.set_default_dtype(torch.float32)
model.eval()
sd = model.state_dict()
B, L = int(os.environ.get("B", "32")), int(os.environ.get("L", "384"))
ids = torch.randint(0, 151643, (B, L), device=dev)
e0 = HipEngine(cfg, sd, dev)
e1 = HipEngine(cfg, sd, dev)
e2 = HipEngine(cfg, sd, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = B // 2


def single():
    e0.forward_hidden(ids)


def dual():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        e1.forward_hidden(ids[:h])
    with torch.cuda.stream(s2):
        e2.forward_hidden(ids[h:])
    cur.wait_stream(s1)
    cur.wait_stream(s2)


ref = e0.forward_hidden(ids).clone()
dual()
torch.cuda.synchronize()
got = torch.cat([e1.hn, e2.hn])
print("dual == single:", torch.equal(ref, got), flush=True)
for rnd in range(4):
    for name, fn in (("single", single), ("dual", dual)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"round {rnd} {name:6s} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms per forward", flush=True)
