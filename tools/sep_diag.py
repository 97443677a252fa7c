#!/usr/bin/env python3
"""Diagnostic: a separated golden case, HIP path vs oracle, step by step (confidences and chosen sets)."""
import ast, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_cpu
from test_hip_e2e import build, SMALL, DEV
from coda_amd import DLMGenerationConfig

z = np.load(os.path.join(ROOT, "tests/golden/separated_vectors.npz"))
d, sd, m = build(SMALL, 22)
for key in sorted(k[:-5] for k in z.files if k.endswith("_meta")):
    if len(sys.argv) > 1 and sys.argv[1] not in key:
        continue
    meta = ast.literal_eval(str(z[key + "_meta"]))
    ids = torch.from_numpy(z[key + "_ids"])
    kw = dict(max_new_tokens=meta["G"], steps=meta["steps"], alg=meta["alg"])
    for rc in ("bf16", None):
        trace = []
        out = m.diffusion_generate(ids.to(DEV), generation_config=DLMGenerationConfig(
            mask_token_id=d.mask_id, pad_token_id=d.pad_id, return_dict_in_generate=True, output_history=True,
            round_conf_to=rc, **kw), debug_trace=trace)
        hist = torch.stack([h.cpu() for h in out.history])
        ref_hist = torch.from_numpy(z[key + "_hist"]).long()
        print(f"== {key} round_conf_to={rc}: steps equal {(hist == ref_hist).all(dim=(1,2)).tolist()}")
    tr = ref_cpu.diffusion_generate(sd, d, ids, trace=True, **kw)
    P = meta["P"]
    for i in range(meta["steps"]):
        ht = trace[i]
        n = int(ht["count"].item())
        dst = ht["dst"][:n].long().cpu()
        print(f" step {i} k={tr.k[i]} same-state={torch.equal(ref_hist[i-1], hist[i-1]) if i else True}")
        cand = tr.cand[i].reshape(-1).nonzero().squeeze(1)
        if not torch.equal(cand, dst):
            print("   candidate sets differ (state diverged)"); break
        ch = ht["conf"].cpu()[dst]; co = tr.conf[i].float()
        xo, xh = tr.x0[i], ht["x0"].cpu()[dst]
        for j in range(n):
            print(f"   pos {int(dst[j])}: conf oracle {co[j]:.5f} hip {ch[j]:.5f} diff {ch[j]-co[j]:+.5f} | x0 oracle {int(xo[j])} hip {int(xh[j])} gap {float(tr.gap[i][j]):.4f}")
        if not torch.equal(ref_hist[i], hist[i]):
            print("   ^ chosen set / tokens differ at this step"); break
