#!/usr/bin/env python3
"""B = 1 serving shape (C1: one sequence, 128 + 256 tokens, 128 steps, block 32; and the FastAPI defaults), one process,
knobs alternated: ms per denoise step with the fused split-K epilogues on / off.  ROUNDS=3 python tools/b1_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_amd import CoDAConfig, CoDALanguageModel  # noqa: E402

dev = torch.device("cuda")
cfg = CoDAConfig()
torch.manual_seed(1)
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    model = CoDALanguageModel(cfg)
torch.set_default_dtype(torch.float32)
model.eval()
with torch.no_grad():
    model.lm_head.weight.mul_(8.0)
eng = model.hip_engine()
if os.environ.get("SPLITK"):      # e.g. SPLITK="2048:2048:4,4096:2048:2,2048:6144:4" = N:K:ranges at M = 384 (and the 32-row compact tail follows)
    from coda_amd import engine as _engine
    for item in os.environ["SPLITK"].split(","):
        n_, k_, s_ = (int(v) for v in item.split(":"))
        _engine.SPLITK_OVERRIDE[(384, n_, k_)] = s_
    print("split-K override:", _engine.SPLITK_OVERRIDE, flush=True)
ids = torch.randint(0, 151643, (1, 128), device=dev)
cases = [("C1 (block 32, greedy)", dict(alg="entropy", block_length=32)),
         ("serving defaults (alg_temp 0.1, no window)", dict(alg="entropy", alg_temp=0.1))]
outs = {}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for fused in (True, False):
        eng.fuse_splitk_epilogues = fused
        for name, kw in cases:
            torch.manual_seed(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model.diffusion_generate(ids, max_new_tokens=256, steps=128, mask_token_id=cfg.mask_token_id, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            outs.setdefault(name, []).append(out.cpu())
            if rnd:
                print(f"round {rnd} fused_epilogues={int(fused)} {name}: {dt / 128 * 1e3:.3f} ms per denoise step", flush=True)
for name, o in outs.items():
    print(name, "all runs token-identical:", all(torch.equal(o[0], x) for x in o[1:]))
