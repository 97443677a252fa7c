"""BASELINE.json configurations C3, C4 and C5 at THEIR OWN shape on the 1.7B architecture (C2 is tests/test_hip_fullsize.py):

  C3  per-GPU shard of the 8-GPU run: B = 64, 512-token prompts, 256 new tokens, 128 steps, block 32, entropy, greedy
  C4  long form: B = 8, 1024-token prompts, 1024 new tokens, 512 steps, block 64, T = 0.7, top-p 0.9, maskgit_plus
  C5  few-step: B = 128, 128-token prompts, 512 new tokens, 32 steps, topk_margin, alg_temp 0.1, no window

Per config: known-answer schedule, window bookkeeping, monotone unmasking, no masks left, determinism
under the same seed, and a bounded TEACHER-FORCED sample against the bf16 CPU oracle.  The stochastic configs run with
the product's own noise (in-kernel counter hash); the oracle replays it from the host restatement of the same hash
(tests/parity.py::hash_uniform_host), keyed by the seed the run used.  Each config's ms per denoise step and
whole-step roofline fraction go to gpurun_out/r03_config_sweep.json (kept as profiles/r03_config_sweep.json)."""
import json
import os
import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from coda_amd.generation_utils import DeviceNoise  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from parity import lm_head_gain  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BIG = os.environ.get("CODA_PARITY_BIG") == "1"     # larger teacher-forced samples (kept log: profiles/r03_parity_big.log)
CONFIGS = {
    "C3": dict(B=64, P=512, G=256, steps=128, block=32, gen=dict(alg="entropy", temperature=0.0), ks=[1] + [2] * 14 + [3],
               sample={8.0: ([0, 9, 18, 27, 36, 45, 54, 63], [0, 1, 15, 16, 31, 32, 47, 48, 63, 64, 79, 80, 95, 111, 112, 127]),
                       1.0: ([4, 23, 41, 60], [0, 16, 63, 127])}),
    "C4": dict(B=8, P=1024, G=1024, steps=512, block=64, gen=dict(alg="maskgit_plus", temperature=0.7, top_p=0.9),
               ks=[1] + [2] * 30 + [3],
               sample={8.0: (list(range(8)), [0, 1, 31, 32, 33, 100, 160, 255, 256, 300, 352, 415, 448, 480, 510, 511]) if BIG else
                            ([1, 3, 4, 6], [0, 31, 32, 100, 255, 256, 480, 511])}),
    "C5": dict(B=128, P=128, G=512, steps=32, block=None, gen=dict(alg="topk_margin", alg_temp=0.1, temperature=0.0),
               ks=[15] + [16] * 30 + [17],
               sample={8.0: ([5, 20, 40, 61, 77, 90, 108, 120], list(range(0, 32, 2))) if BIG else ([5, 40, 77, 120], [0, 1, 8, 15, 16, 24, 30, 31]),
                       1.0: ([33, 99], [0, 12, 24, 31])}),
}
_RUNS = {}


def per_step_flops(B, L, rows):
    """forward over B*L tokens + lm-head over the consumed rows."""
    return B * L * (2_818_572_288 + 229_376 * L) + rows * 622_329_856


def run_config(full_model, name, gain=8.0):
    if (name, gain) in _RUNS:
        return _RUNS[(name, gain)]
    cfg, m = full_model
    c = CONFIGS[name]
    B, P, G = c["B"], c["P"], c["G"]
    g = torch.Generator().manual_seed(1000 + int(name[1]))                 # seed = 1000 + config id
    ids = torch.randint(0, 151643, (B, P), generator=g)
    kw = dict(max_new_tokens=G, steps=c["steps"], block_length=c["block"], mask_token_id=cfg.mask_token_id, **c["gen"])
    torch.manual_seed(77)
    noise = DeviceNoise()                                                  # the product's own noise source, seed known
    trace = []
    with lm_head_gain(m, gain):
        out = m.diffusion_generate(ids.to(DEV), return_dict_in_generate=True, output_history=True, debug_trace=trace,
                                   noise_source=noise, **kw)
        hist = [h.cpu() for h in out.history]
        # timing: the same call without history / trace, twice (warm), same seed -> also the determinism check
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        again = m.diffusion_generate(ids.to(DEV), noise_source=noise, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    r = dict(cfg=cfg, m=m, ids=ids, kw=kw, noise=noise, seq=out.sequences.cpu(), hist=hist, trace=trace, again=again.cpu(), seconds=dt)
    _RUNS[(name, gain)] = r
    return r


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_schedule_bookkeeping_determinism(full_model, name):
    c = CONFIGS[name]
    r = run_config(full_model, name)
    cfg, ids, seq, hist, trace = r["cfg"], r["ids"], r["seq"], r["hist"], r["trace"]
    B, P, G, S = c["B"], c["P"], c["G"], c["steps"]
    L, mask = P + G, cfg.mask_token_id
    bl = c["block"] or G
    steps_b = S // (G // bl)
    assert c["ks"] == ref_cpu.transfer_counts(bl, 1, steps_b, 1e-3)        # known answers
    assert seq.shape == (B, L) and len(hist) == S and torch.equal(hist[-1], seq)
    assert torch.equal(seq[:, :P], ids)
    assert (seq != mask).all() and (seq >= 0).all() and (seq < cfg.vocab_size).all()
    prev = F.pad(ids, (0, G), value=mask)
    pos = torch.arange(L)[None, :]
    consumed = 0
    for i, cur in enumerate(hist):
        blk, j = divmod(i, steps_b)
        lo, hi = P + blk * bl, P + (blk + 1) * bl
        changed = cur != prev
        assert not changed[:, :lo].any() and not changed[:, hi:].any()            # only the current window moves
        assert (prev[changed] == mask).all() and (cur[changed] != mask).all()     # masked -> token, never back
        assert (changed.sum(1) == c["ks"][j]).all(), (name, i, changed.sum(1).tolist()[:8])
        t = trace[i]
        n = int(t["count"].item())
        want = ((prev == mask) & (pos >= lo) & (pos < hi)).reshape(-1).nonzero().squeeze(1)
        assert n == want.numel() and torch.equal(t["dst"][:n].long().cpu(), want)
        consumed += n
        prev = cur
    # determinism: the same seed gives the same tokens (the noise is a pure function of seed, step, position, index)
    assert torch.equal(r["again"], seq), name
    # measurement record
    ms = r["seconds"] / S * 1e3
    flops = S * per_step_flops(B, L, 0) + consumed * 622_329_856
    rec = dict(config=name, B=B, L=L, steps=S, block_length=c["block"], **{k: v for k, v in c["gen"].items()},
               ms_per_denoise_step=round(ms, 3), generated_tokens_per_s=round(B * G / r["seconds"], 1),
               whole_step_tflops_survey_formula=round(flops / r["seconds"] / 1e12, 1),
               frac_of_bf16_mfma_peak=round(flops / r["seconds"] / 1e12 / 2500.0, 4), lm_head_rows_consumed=consumed)
    print(f"{name}: {rec}")
    path = os.path.join(ROOT, "gpurun_out", "r03_config_sweep.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = rec
    json.dump(data, open(path, "w"), indent=1)


@pytest.mark.parametrize("name,gain", [(n, g) for n in CONFIGS for g in CONFIGS[n]["sample"]])
def test_config_teacher_forced_sample_vs_cpu_oracle(full_model, name, gain):
    """``gain`` 8: the bench's lm_head x 8; 1: a second run of the configuration on the unscaled lm_head (|logit| < 8 as in a
    trained checkpoint), where one bf16 ulp of a logit is 2^-5 and the confidence tolerance is tight."""
    from parity import FWD_ERR_ULPS_FULL, assert_parity, describe, hash_uniform_host, subsample, teacher_forced
    c = CONFIGS[name]
    r = run_config(full_model, name, gain)
    cfg, m, ids, hist, trace, noise = r["cfg"], r["m"], r["ids"], r["hist"], r["trace"], r["noise"]
    B, P, G = c["B"], c["P"], c["G"]
    L, V = P + G, cfg.vocab_size
    assert (r["seq"] != cfg.mask_token_id).all() and torch.equal(r["again"], r["seq"])
    d = ref_cpu.Dims()
    with lm_head_gain(m, gain):
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    rows, steps = c["sample"][gain]
    x_init = F.pad(ids, (0, G), value=cfg.mask_token_id)
    states, sub_hist, sub_trace = subsample(x_init, hist, trace, rows, steps)
    # the run's noise, regenerated on the host for the sampled rows / steps: the kernels key it by the FLAT position in
    # the full batch (orig_dst), the oracle wants it per candidate row (token race) and per [row, position] (selection)
    log = []
    gen = c["gen"]
    for n, i in enumerate(steps):
        seed = noise.step_seed(i)
        if gen.get("temperature", 0.0) > 0:
            key = sub_trace[n]["orig_dst"]
            u = hash_uniform_host(seed, key[:, None], torch.arange(V)[None, :])
            log.append(("token", i, -torch.log(u)))
        if gen.get("alg_temp"):
            flat = torch.tensor(rows)[:, None] * L + torch.arange(L)[None, :]
            log.append(("select", i, -torch.log(hash_uniform_host(seed, flat, 0xFFFFFFFF))))
    stats, tr = teacher_forced(sd, d, ids[rows], sub_hist, sub_trace, noise_log=log or None, states=states, step_indices=steps,
                               fwd_err_ulps=FWD_ERR_ULPS_FULL, max_new_tokens=G, steps=c["steps"], block_length=c["block"], **gen)
    print(f"{name} at full size, lm_head x {gain:g}, teacher-forced sample ({len(rows)} rows x {len(steps)} steps): {describe(stats)}")
    # T = 0.7 on |logit| ~ 40: two logits 8 ulps apart differ by e^2.9 in probability - on the x 8 head the sampled-token
    # band excuses most of what differs, the tight evidence for that path is the exact-logit kernel test
    assert_parity(stats, max_excused=0.30 if gen.get("temperature", 0.0) > 0 else 0.12, min_candidates=64,
                  max_wide=0.15 if gen.get("temperature", 0.0) > 0 else 0.03)
