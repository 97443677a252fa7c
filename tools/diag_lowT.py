#!/usr/bin/env python3
"""Diagnostic for T = 0.1 + top-p token mismatches (tests/test_hip_e2e.py::test_lm_eval_adapter_left_padded[0.1]):
for every candidate whose token differs from the oracle's, prints both paths' logits of both tokens, the noise, and
whether the other path's token survived the ORACLE's top-p filter."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_cpu  # noqa: E402
from test_hip_e2e import SMALL, build, Replay  # noqa: E402
from coda_amd.lm_eval_adapter import DiffLLMAdapter  # noqa: E402
from parity import teacher_forced, token_band  # noqa: E402

d, sd, m = build(SMALL, 22)
T = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
g = torch.Generator().manual_seed(77)
prompts = [torch.randint(0, d.pad_id, (n,), generator=g).tolist() for n in (9, 4, 12, 1)]
a = DiffLLMAdapter(m, batch_size=4, max_prompt_len=10, max_new_tokens=16, diffusion_steps=16, temperature=T, top_p=0.9)
ids, attn = a.prepare_batch(prompts)
kw = dict(max_new_tokens=16, steps=16, alg="entropy", temperature=T, top_p=0.9, alg_temp=0.0)
torch.manual_seed(321)
noise = ref_cpu.TorchNoise()
ref_cpu.diffusion_generate(sd, d, ids.cpu(), noise=noise, **kw)
gk = a.generation_kwargs()
gk.update(output_history=True)
trace = []
out = m.diffusion_generate(ids, attention_mask=attn, debug_trace=trace, noise_source=Replay(noise.log, d.vocab), **gk)
hist = [h.cpu() for h in out.history]
stats, tr = teacher_forced(sd, d, ids.cpu(), hist, trace, noise_log=noise.log, **kw)
print(stats["violations"])
x_init = F.pad(ids.cpu(), (0, 16), value=d.mask_id)
states = [x_init] + hist[:-1]
qlog = {s: t for k_, s, t in noise.log if k_ == "token"}
for i in range(len(hist)):
    cand = tr.cand[i]
    flat = cand.reshape(-1).nonzero().squeeze(1)
    x0_h = trace[i]["x0"].cpu()[flat]
    x0_o = tr.x0[i]
    bad = (x0_h != x0_o).nonzero().squeeze(1)
    if not len(bad):
        continue
    with torch.no_grad():
        lo = ref_cpu.forward_logits(sd, d, states[i])
        lo = torch.cat([lo[:, :1], lo[:, :-1]], 1).reshape(-1, d.vocab)[flat]
        lh, _ = m(states[i].to("cuda"))
        lh = torch.cat([lh[:, :1], lh[:, :-1]], 1).reshape(-1, d.vocab).cpu()[flat]
    band = token_band(tr.absmax[i], T)
    for r in bad.tolist():
        o, h = int(x0_o[r]), int(x0_h[r])
        row_o = (lo[r] / T)
        filt = ref_cpu.top_p_filter(row_o[None], 0.9)[0]
        kept_o = filt > torch.finfo(filt.dtype).min
        row_h = (lh[r] / T)
        filt_h = ref_cpu.top_p_filter(row_h[None], 0.9)[0]
        kept_h = filt_h > torch.finfo(filt_h.dtype).min
        q = qlog[i][r]
        po = torch.softmax(filt.float(), -1)
        print(f"step {i} cand {r}: oracle tok {o} hip tok {h} | oracle logits {float(lo[r, o]):.4f} {float(lo[r, h]):.4f} | "
              f"hip logits {float(lh[r, o]):.4f} {float(lh[r, h]):.4f} | q {float(q[o]):.4g} {float(q[h]):.4g} | "
              f"oracle keeps (its own logits) {int(kept_o.sum())} tokens, hip's token kept by oracle: {bool(kept_o[h])}; "
              f"oracle top-p on HIP's logits keeps {int(kept_h.sum())}, oracle's token kept there: {bool(kept_h[o])} | "
              f"gap {float(tr.gap[i][r]):.4f} band {float(band[r]):.4f} absmax {float(tr.absmax[i][r]):.2f} "
              f"max |dlogit| row {float((lo[r].float() - lh[r].float()).abs().max()):.4f}")
