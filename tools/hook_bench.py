#!/usr/bin/env python3
"""Hooks at the C2 shape (batch 32, 128 + 256 tokens, block 32, entropy, greedy), 32 of the 128 denoise steps, one process:
no hook / a tokens hook that does not read its logits argument (LazyLogits stays unmaterialised: fused route) / a tokens
hook that reads it every step / an identity logits hook (shifted [B, L, V] logits built every step)."""
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
ids = torch.randint(0, 151643, (32, 128 + 192), device=dev)          # 2 of 8 blocks: the same L = 384
kw = dict(max_new_tokens=64, steps=32, block_length=32, alg="entropy", mask_token_id=cfg.mask_token_id)
peek = []
cases = [("no hook", {}),
         ("tokens hook, logits not read", dict(generation_tokens_hook_func=lambda s, x, l: x)),
         ("tokens hook reading logits[:, -1]", dict(generation_tokens_hook_func=lambda s, x, l: (peek.append(None if l is None else l[:, -1].float().sum()), x)[1])),
         ("identity logits hook", dict(generation_logits_hook_func=lambda s, x, l: l))]
outs = {}
for rnd in range(3):
    for name, extra in cases:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.diffusion_generate(ids, **kw, **extra)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        outs[name] = out
        if rnd:
            print(f"round {rnd} {name:36s}: {dt / 32 * 1e3:8.2f} ms per denoise step", flush=True)
base = outs["no hook"]
for name, o in outs.items():
    print(f"{name:36s}: tokens equal to the hook-free run: {float((o == base).float().mean()):.4f}")
