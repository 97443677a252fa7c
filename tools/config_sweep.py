#!/usr/bin/env python3
"""Runs a few denoise steps of every BASELINE.json config shape at full model size (functional + ms/step)."""
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

# name, B, P, G, total steps, block, extra kwargs, steps actually run (a whole number of blocks)
CONFIGS = [
    # B = 1: the reference's CPU-runnable case on the GPU, and the FastAPI service's defaults (config.py:35-42: 256 tokens,
    # 128 steps, entropy, alg_temp 0.1; block_length read but never passed on - SURVEY F2).  Weight-stream floor:
    # 4.06 GB of bf16 parameters / 8 TB/s = 0.51 ms per denoise step
    ("C1 on GPU (B=1)", 1, 128, 256, 128, 32, dict(alg="entropy"), 32),
    ("serving default (B=1)", 1, 128, 256, 128, None, dict(alg="entropy", alg_temp=0.1), 16),
    ("C2 window", 32, 128, 256, 128, 32, dict(alg="entropy"), 16),
    ("C2 reference", 32, 128, 256, 128, None, dict(alg="entropy"), 8),
    ("C3/GPU window", 64, 512, 256, 128, 32, dict(alg="entropy"), 16),
    ("C4 long", 8, 1024, 1024, 512, 64, dict(alg="maskgit_plus", temperature=0.7, top_p=0.9), 32),
    ("C5 few-step", 128, 128, 512, 32, None, dict(alg="topk_margin", alg_temp=0.1), 4),
    # the lm-eval scripts' workload (eval_mbpp_humaneval.sh: batch 32, 768 new tokens, 768 steps, T=0.1,
    # top_p=0.9, entropy); the first steps (every generated position still masked) are the expensive ones
    ("E1 lm-eval", 32, 128, 768, 768, None, dict(alg="entropy", temperature=0.1, top_p=0.9, alg_temp=0.0), 8),
]
if os.environ.get("FUSED") == "1":              # A/B: the fused lm-head kernel at every row count (engine.materialise_min_rows)
    model.hip_engine().materialise_min_rows = 1 << 30
only = sys.argv[1:]
for name, B, P, G, steps, bl, kw, run in CONFIGS:
    if only and not any(o in name for o in only):
        continue
    ids = torch.randint(0, 151643, (B, P), device=dev)
    # run only `run` steps: shrink G/steps proportionally in whole blocks (window mode) or cut steps (reference mode)
    if bl is not None:
        nb = max(1, run // (steps // (G // bl)))
        g_run, s_run = nb * bl, nb * (steps // (G // bl))
        pad = G - g_run           # keep L identical: extra prompt tokens stand in for the untouched blocks
        ids2 = torch.cat([ids, torch.randint(0, 151643, (B, pad), device=dev)], dim=1) if pad else ids
    else:
        g_run, s_run, ids2 = G, run, ids
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        out = model.diffusion_generate(ids2, max_new_tokens=g_run, steps=s_run, block_length=bl, mask_token_id=cfg.mask_token_id, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    left = int((out == cfg.mask_token_id).sum())
    print(f"{name:14s} B={B} L={P+G} steps_run={s_run}: {dt/s_run*1e3:8.2f} ms/denoise step, masks left {left}, "
          f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
