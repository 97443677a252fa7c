"""Probe behind tests/test_hip_fullsize.py::test_c2_full_size_forward_logits_vs_cpu_oracle: distribution of the logit
deviations HIP-bf16 / oracle-bf16 / oracle-fp32 at full size (28 layers, V = 151 936), lm_head x 8 and x 1."""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from coda_amd import CoDAConfig, CoDALanguageModel
from oracle import ref_cpu
from parity import bf16_ulp

cfg = CoDAConfig()
torch.manual_seed(4321)
torch.set_default_dtype(torch.bfloat16)
with torch.device("cuda"):
    m = CoDALanguageModel(cfg)
torch.set_default_dtype(torch.float32)
m.eval()
d = ref_cpu.Dims()
g = torch.Generator().manual_seed(1002)
ids = torch.randint(0, 151643, (3, 128), generator=g)
x = F.pad(ids, (0, 256), value=cfg.mask_token_id)
x[1, 128:256] = torch.randint(0, 151643, (128,), generator=g)
x[2, 128:] = torch.randint(0, 151643, (256,), generator=g)

def q(t, qs=(0.5, 0.9, 0.99, 0.999, 0.9999)):
    f = t.reshape(-1)
    if f.numel() > 20_000_000:
        f = f[torch.randint(0, f.numel(), (20_000_000,))]
    return " ".join(f"p{100*p:g}={float(torch.quantile(f, p)) if f.numel() < 16_000_000 else float(torch.kthvalue(f, int(p*f.numel()))[0]):.3f}" for p in qs) + f" max={float(t.max()):.3f} mean={float(t.mean()):.3f}"

for gain in (8.0, 1.0):
    with torch.no_grad():
        m.lm_head.weight.mul_(gain)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    sd32 = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        hip = m.hip_engine().logits(x.cuda()).float().cpu()
        t0 = time.perf_counter(); ref = ref_cpu.forward_logits(sd, d, x).float(); t1 = time.perf_counter()
        tru = ref_cpu.forward_logits(sd32, d, x); t2 = time.perf_counter()
    print(f"== lm_head x {gain:g}: oracle bf16 {t1-t0:.1f} s, fp32 {t2-t1:.1f} s for 3 x 384 tokens; logit std {float(tru.std()):.3f}")
    ulp = bf16_ulp(tru.abs().amax(-1, keepdim=True))
    top = tru >= tru.amax(-1, keepdim=True) - 16 * ulp
    for name, a, b in (("hip  vs bf16 oracle", hip, ref), ("hip  vs fp32 truth ", hip, tru), ("bf16 oracle vs fp32", ref, tru)):
        e = (a - b).abs() / ulp
        print(f"{name}: all  {q(e)}")
        print(f"{name}: top  {q(e[top])}  (n={int(top.sum())})")
        print(f"{name}: row-rms {q(e.pow(2).mean(-1).sqrt())}")
    am = lambda t: t.argmax(-1)
    print(f"argmax agreement: hip/bf16 {float((am(hip)==am(ref)).float().mean()):.4f} hip/fp32 {float((am(hip)==am(tru)).float().mean()):.4f} bf16/fp32 {float((am(ref)==am(tru)).float().mean()):.4f}")
    # per-position structure: which rows carry the large errors
    e = ((hip - ref).abs() / ulp).pow(2).mean(-1).sqrt()
    e2 = ((ref - tru).abs() / ulp).pow(2).mean(-1).sqrt()
    for s in range(3):
        print(f"seq {s}: row-rms hip/bf16 prompt {float(e[s,:128].mean()):.3f} gen {float(e[s,128:].mean()):.3f} | bf16/fp32 prompt {float(e2[s,:128].mean()):.3f} gen {float(e2[s,128:].mean()):.3f}")
    with torch.no_grad():
        m.lm_head.weight.mul_(1.0 / gain)
