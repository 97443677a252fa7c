"""Full-size checks: the 1.7B architecture (28 layers, V = 151 936) on the BASELINE.json C2 workload
(batch 32, 128-token prompts, 256 new tokens, 128 denoise steps, block_length 32, entropy, greedy).

The CPU oracle needs ~1 s per sequence and forward at this size, so parity is checked two ways:
  * size-independent properties of the whole run (SURVEY.md 8a known-answer schedules, window
    bookkeeping, monotone unmasking, determinism, row independence);
  * a bounded TEACHER-FORCED sample: two sequences x seven denoise steps of the same run are
    replayed through the bf16 CPU oracle from the HIP path's own states (tests/parity.py bands).
Weights are random-init (reference init semantics, lm_head x 8 as in bench.py); no checkpoint ships
with the reference."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BIG = os.environ.get("CODA_PARITY_BIG") == "1"     # larger teacher-forced samples (kept log: profiles/r03_parity_big.log)

from coda_amd import CoDAConfig, CoDALanguageModel, DLMGenerationConfig  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from parity import lm_head_gain  # noqa: E402

DEV = "cuda"
B, P, G, STEPS, BL = 32, 128, 256, 128, 32
L = P + G


def run_c2(cfg, m, ids):
    trace = []
    out = m.diffusion_generate(ids.to(DEV), generation_config=DLMGenerationConfig(
        mask_token_id=cfg.mask_token_id, return_dict_in_generate=True, output_history=True, max_new_tokens=G, steps=STEPS,
        alg="entropy", temperature=0.0, block_length=BL), debug_trace=trace)
    return out.sequences.cpu(), [h.cpu() for h in out.history], trace


@pytest.fixture(scope="module")
def full(full_model):
    cfg, m = full_model
    g = torch.Generator().manual_seed(1002)
    ids = torch.randint(0, 151643, (B, P), generator=g)
    seq, hist, trace = run_c2(cfg, m, ids)
    return cfg, m, ids, seq, hist, trace


def test_c2_schedule_and_window_bookkeeping(full):
    cfg, m, ids, seq, hist, trace = full
    mask = cfg.mask_token_id
    assert seq.shape == (B, L) and seq.dtype == torch.int64 and len(hist) == STEPS
    assert torch.equal(seq[:, :P], ids)                                   # prompt untouched
    assert (seq != mask).all() and (seq >= 0).all() and (seq < cfg.vocab_size).all()
    assert torch.equal(hist[-1], seq)
    # known answers: block 32 / 16 steps per block reveals 1, 2 x 14, 3 positions per row
    ks = [1] + [2] * 14 + [3]
    assert ks == ref_cpu.transfer_counts(BL, 1, STEPS // (G // BL), 1e-3)
    prev = F.pad(ids, (0, G), value=mask)
    for i, cur in enumerate(hist):
        blk, j = divmod(i, STEPS // (G // BL))
        lo, hi = P + blk * BL, P + (blk + 1) * BL
        changed = cur != prev
        assert not changed[:, :lo].any() and not changed[:, hi:].any()    # only the current window moves
        assert (prev[changed] == mask).all() and (cur[changed] != mask).all()   # masked -> token, never back
        assert (changed.sum(1) == ks[j]).all(), (i, changed.sum(1).tolist())
        assert ((cur[:, hi:] == mask).all() and (cur[:, :lo] != mask).all())
        # the trace's candidate list is exactly the window's masked positions, row-major
        t = trace[i]
        n = int(t["count"].item())
        want = ((prev == mask) & (torch.arange(L)[None, :] >= lo) & (torch.arange(L)[None, :] < hi)).reshape(-1).nonzero().squeeze(1)
        assert n == want.numel() and torch.equal(t["dst"][:n].long().cpu(), want)
        prev = cur


def test_c2_deterministic_and_row_independent(full):
    cfg, m, ids, seq, hist, trace = full
    kw = dict(max_new_tokens=G, steps=STEPS, alg="entropy", temperature=0.0, block_length=BL, mask_token_id=cfg.mask_token_id)
    again = m.diffusion_generate(ids.to(DEV), **kw).cpu()
    assert torch.equal(again, seq)
    # Rows do not interact.  A 4-row shard takes other code paths than the 32-row batch (T = 1536 tokens:
    # unfused QKV + standalone norm/rotary kernel, split-K in the compact tail, other lm-head split geometry), i.e. other fp32
    # summation orders, so its bf16 values differ from the batch's by an ulp here and there; free running, the first
    # near-tie that flips then changes every later step (F5), which says nothing about coupling.  So the check is ONE
    # denoise step from the same state: hidden states within a few bf16 ulps, confidences close, tokens equal except on
    # near-ties.
    eng = m.hip_engine()
    rows = slice(8, 12)
    step = 40
    blk = step // (STEPS // (G // BL))
    lo, hi = P + blk * BL, P + (blk + 1) * BL
    out = {}
    for name, xs in (("batch", hist[step - 1].to(DEV)), ("shard", hist[step - 1][rows].to(DEV))):
        eng.set_mask_id(cfg.mask_token_id)
        hn = eng.forward_hidden(xs, need=(lo - 1, hi - 1)).view(xs.shape[0], L, -1)[:, lo - 1:hi - 1].float().clone()
        eng.build_candidates(xs, lo, hi)
        eng.lmhead_confidence(xs.shape[0] * BL, 2, 0.0, None, 0, False)
        torch.cuda.synchronize()
        n = int(eng.count.item())
        dst = eng.dst_pos[:n].long()
        out[name] = (hn, dst.cpu(), eng.conf[dst].cpu().clone(), eng.x0[dst].cpu().clone())
    hb, db, cb, xb = out["batch"]
    hs, ds, cs, xs_ = out["shard"]
    hb = hb[rows]
    sel = (db // L >= 8) & (db // L < 12)
    assert torch.equal(db[sel] - 8 * L, ds)                               # same candidate positions
    dh = (hb - hs).abs()
    tok_same = (xb[sel] == xs_).float().mean().item()
    dc = (cb[sel] - cs).abs()
    print(f"row independence, one step from the same state (4-row shard vs the same rows of the 32-row batch): hidden max |diff| "
          f"{dh.max():.3g} mean {dh.mean():.3g} (mean |h| {hb.abs().mean():.3g}); {ds.numel()} candidates, tokens equal {tok_same:.3f}, "
          f"max |conf diff| {dc.max():.3g}")
    # measured: mean |diff| 1.6 % of mean |h| (the same size as the HIP-vs-CPU-oracle deviation of this random-init 28-layer
    # model: one-ulp differences grow through the layers), 61 of 68 tokens equal, the others near-ties
    assert dh.mean().item() < 0.04 * hb.abs().mean().item() and dh.max().item() < 0.5
    assert tok_same >= 0.8 and dc.median().item() < 0.05


def test_c2_last_layer_compact_tail_equals_full_forward(full):
    """The last decoder layer's per-token tail runs on the consumed rows only (engine.forward_hidden(need=...)); those
    rows must equal the full forward's, and a generation with the compaction switched off must give the same tokens."""
    cfg, m, ids, seq, hist, trace = full
    eng = m.hip_engine()
    assert eng.prune_last_layer
    x = hist[40].to(DEV)
    lo, hi = P + 2 * BL - 1, P + 3 * BL - 1                                # source rows of block 2's window
    full_hn = eng.forward_hidden(x).view(B, L, -1)[:, lo:hi].clone()
    part_hn = eng.forward_hidden(x, need=(lo, hi)).view(B, L, -1)[:, lo:hi].clone()
    diff = (full_hn.float() - part_hn.float()).abs().max().item()
    print(f"compact tail vs full forward: max |diff| of the final-normed rows {diff:.3g} (scale {full_hn.float().abs().mean().item():.3g})")
    assert torch.equal(full_hn, part_hn)
    kw = dict(mask_token_id=cfg.mask_token_id, max_new_tokens=64, steps=32, alg="entropy", temperature=0.0, block_length=BL)
    a = m.diffusion_generate(ids[:8].to(DEV), **kw)
    eng.prune_last_layer = False
    try:
        b = m.diffusion_generate(ids[:8].to(DEV), **kw)
    finally:
        eng.prune_last_layer = True
    assert torch.equal(a, b)


C2_SAMPLES = {   # lm_head gain -> (rows, steps): block starts / ends, middles, the last step
    8.0: ([0, 5, 9, 13, 17, 21, 26, 31], [0, 1, 15, 16, 17, 31, 32, 47, 48, 63, 64, 80, 96, 111, 112, 127]),
    1.0: ([2, 11, 20, 29], [0, 15, 16, 40, 63, 64, 100, 127]),
}


@pytest.mark.parametrize("gain", [8.0, 1.0])
def test_c2_teacher_forced_sample_vs_cpu_oracle(full, gain):
    """C2 at its own shape, teacher-forced against the bf16 CPU oracle: 8 rows x 16 steps on the bench's lm_head x 8, and
    4 rows x 8 steps of a second run on the UNSCALED lm_head (|logit| < 8 as in a trained checkpoint: one bf16 ulp of a logit
    is 2^-5, the confidence tolerance drops below 0.2 and the check means something).  On the unscaled run the confidences
    are also compared THREE ways against the fp32 oracle: the HIP path must sit as close to exact arithmetic as the
    reference's own bf16 run does."""
    from parity import FWD_ERR_ULPS_FULL, assert_parity, describe, subsample, teacher_forced
    cfg, m, ids, seq, hist, trace = full
    d = ref_cpu.Dims()
    assert (d.vocab, d.hidden, d.layers, d.mask_id) == (cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.mask_token_id)
    rows, steps = C2_SAMPLES[gain]
    with lm_head_gain(m, gain):
        if gain != 8.0:
            seq, hist, trace = run_c2(cfg, m, ids)
            assert (seq != cfg.mask_token_id).all()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x_init = F.pad(ids, (0, G), value=cfg.mask_token_id)
    states, sub_hist, sub_trace = subsample(x_init, hist, trace, rows, steps)
    kw = dict(max_new_tokens=G, steps=STEPS, alg="entropy", block_length=BL)
    stats, tr = teacher_forced(sd, d, ids[rows], sub_hist, sub_trace, states=states, step_indices=steps,
                               fwd_err_ulps=FWD_ERR_ULPS_FULL, **kw)
    print(f"C2 full size, lm_head x {gain:g}, teacher-forced sample ({len(rows)} rows x {len(steps)} steps): {describe(stats)}")
    assert_parity(stats, max_excused=0.12, min_candidates=400)
    if gain == 1.0:
        # noise floor of the confidences: fp32 oracle (exact arithmetic for this purpose) on a part of the same sample
        n_s = 4
        sd32 = {k: v.float() for k, v in sd.items()}
        with torch.no_grad():
            t32 = ref_cpu.diffusion_generate(sd32, d, ids[rows], force_states=states[:n_s], trace=True, step_indices=steps[:n_s], **kw)
        e_hip, e_ref = [], []
        for i in range(n_s):
            flat = tr.cand[i].reshape(-1).nonzero().squeeze(1)
            same = (tr.x0[i] == t32.x0[i]) & (sub_trace[i]["x0"][flat] == t32.x0[i])
            e_hip.append((sub_trace[i]["conf"][flat] - t32.conf[i]).abs()[same])
            e_ref.append((tr.conf[i].float() - t32.conf[i]).abs()[same])
        e_hip, e_ref = torch.cat(e_hip), torch.cat(e_ref)
        print(f"   confidences against the fp32 oracle ({e_hip.numel()} candidates): HIP mean |err| {float(e_hip.mean()):.4g} max "
              f"{float(e_hip.max()):.4g}; bf16 oracle mean |err| {float(e_ref.mean()):.4g} max {float(e_ref.max()):.4g}")
        assert e_hip.numel() > 100
        assert float(e_hip.mean()) <= 1.25 * float(e_ref.mean()) + 1e-3 and float(e_hip.max()) <= 1.5 * float(e_ref.max()) + 1e-2


def test_c1_shape_all_steps_teacher_forced(full):
    """C1's shape on the HIP path (BASELINE.json configs[0] / the FastAPI defaults, serving/fast-api/app/core/config.py:35-42):
    batch 1, 128 + 256 tokens, 128 steps, block 32, entropy, greedy - T = 384 tokens, i.e. the split-K GEMM route, the
    unfused QKV + standalone norm / rotary kernel and the compact tail at 32 rows.  EVERY one of the 128 denoise steps is
    replayed through the bf16 CPU oracle."""
    from parity import FWD_ERR_ULPS_FULL, assert_parity, describe, teacher_forced
    cfg, m, ids, _, _, _ = full
    d = ref_cpu.Dims()
    one = ids[7:8]
    trace = []
    kw = dict(max_new_tokens=G, steps=STEPS, alg="entropy", block_length=BL)
    out = m.diffusion_generate(one.to(DEV), mask_token_id=cfg.mask_token_id, return_dict_in_generate=True, output_history=True,
                               debug_trace=trace, temperature=0.0, **kw)
    hist = [h.cpu() for h in out.history]
    assert len(hist) == STEPS and (out.sequences != cfg.mask_token_id).all()
    ks = [1] + [2] * 14 + [3]
    prev = F.pad(one, (0, G), value=cfg.mask_token_id)
    for i, cur in enumerate(hist):
        assert int((cur != prev).sum()) == ks[i % 16], i
        prev = cur
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    stats, tr = teacher_forced(sd, d, one, hist, trace, fwd_err_ulps=FWD_ERR_ULPS_FULL, **kw)
    print(f"C1 shape (B = 1) on the HIP path, all {STEPS} steps teacher-forced: {describe(stats)}")
    assert_parity(stats, max_excused=0.12, min_candidates=2000)


@pytest.mark.parametrize("name,cfg,block", [
    ("C4-like", dict(alg="maskgit_plus", temperature=0.7, top_p=0.9), 16),
    ("C5-like", dict(alg="topk_margin", alg_temp=0.1), None),
    ("E1-like", dict(alg="entropy", temperature=0.1, top_p=0.9, alg_temp=0.0), None),
    ("serving defaults", dict(alg="entropy", alg_temp=0.1), None),
])
def test_full_architecture_stochastic_configs_teacher_forced(full, name, cfg, block):
    """The other BASELINE.json configurations' sampler settings (C4: T = 0.7 + top-p 0.9 + maskgit_plus + windows,
    C5: topk_margin + alg_temp 0.1, the lm-eval scripts' T = 0.1 + top-p, the serving defaults) on the full
    architecture (28 layers, V = 151 936) at a shape the CPU oracle finishes in seconds: the oracle's noise is
    replayed into the HIP kernels and every step is compared teacher-forced."""
    from parity import describe, teacher_forced
    from test_hip_e2e import Replay
    cfg_m, m, _, _, _, _ = full
    d = ref_cpu.Dims()
    # T = 0.1 amplifies a bf16 logit ulp ten times: with the lm_head x 8 "peaked" weights (|logit| ~ 60, ulp 0.25) the
    # honest band would be e^13, i.e. no check at all.  This case runs on the unscaled lm_head (|logit| < 8; the
    # division by 8 is exact in bf16 and is undone afterwards), where the band is that of a real checkpoint's logits.
    rescale = name == "E1-like"
    if rescale:
        with torch.no_grad():
            m.lm_head.weight.mul_(0.125)
    try:
        _stochastic_case(cfg_m, m, d, name, cfg, block)
    finally:
        if rescale:
            with torch.no_grad():
                m.lm_head.weight.mul_(8.0)


def _stochastic_case(cfg_m, m, d, name, cfg, block):
    from parity import FWD_ERR_ULPS_FULL, assert_parity, describe, teacher_forced
    from test_hip_e2e import Replay
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(99)
    ids = torch.randint(0, 151643, (2, 24), generator=g)
    kw = dict(max_new_tokens=32, steps=8, block_length=block, **cfg)
    torch.manual_seed(321)
    noise = ref_cpu.TorchNoise()
    ref = ref_cpu.diffusion_generate(sd, d, ids, noise=noise, **kw)
    trace = []
    out = m.diffusion_generate(ids.to(DEV), mask_token_id=cfg_m.mask_token_id, noise_source=Replay(noise.log, d.vocab),
                               return_dict_in_generate=True, output_history=True, debug_trace=trace, **kw)
    hist = [h.cpu() for h in out.history]
    stats, tr = teacher_forced(sd, d, ids, hist, trace, noise_log=noise.log, fwd_err_ulps=FWD_ERR_ULPS_FULL, **kw)
    rate = (out.sequences.cpu() == ref.sequences).float().mean().item()
    print(f"full architecture, {name}: free-running token match {rate:.3f}; teacher-forced {describe(stats)}")
    assert (out.sequences != cfg_m.mask_token_id).all()
    assert_parity(stats, max_excused=0.20, max_wide=0.08)


def _ulps_stats(a, b, ulp, top):
    err = (a - b).abs() / ulp
    flat = err.reshape(-1)
    kth = lambda t, q: float(torch.kthvalue(t, max(1, int(q * t.numel())))[0])
    et = err[top]
    return dict(mean=float(err.mean()), p999=kth(flat, 0.999), max=float(err.max()),
                top_mean=float(et.mean()), top_p999=kth(et, 0.999), top_max=float(et.max()),
                row_rms_max=float(err.pow(2).mean(-1).sqrt().max()))


def forward_error_stats(m, sd, d, x):
    """Logit deviations of the full 28-layer forward + lm_head, three ways - HIP (bf16) / CPU oracle in bf16 (= the
    reference's arithmetic) / CPU oracle in fp32 (exact arithmetic for this purpose) - per logit in bf16 ulps OF THE ROW'S
    LARGEST |logit| (the unit tests/parity.py states its bands in)."""
    from parity import bf16_ulp
    eng = m.hip_engine()
    with torch.no_grad():
        hip = eng.logits(x.to(DEV)).float().cpu()                          # [k, L, V]
        ref = ref_cpu.forward_logits(sd, d, x).float()
        tru = ref_cpu.forward_logits({k: v.float() for k, v in sd.items()}, d, x)
    ulp = bf16_ulp(tru.abs().amax(-1, keepdim=True))
    top = tru >= tru.amax(-1, keepdim=True) - 16 * ulp                      # entries that can take part in a top-2 decision
    am = lambda t: t.argmax(-1)
    return dict(rows=int(x.numel()), logit_std=float(tru.std()), absmax=float(tru.abs().max()), top_count=int(top.sum()),
                hip_ref=_ulps_stats(hip, ref, ulp, top), hip_tru=_ulps_stats(hip, tru, ulp, top), ref_tru=_ulps_stats(ref, tru, ulp, top),
                agree_hip_ref=float((am(hip) == am(ref)).float().mean()), agree_hip_tru=float((am(hip) == am(tru)).float().mean()),
                agree_ref_tru=float((am(ref) == am(tru)).float().mean()))


def describe_forward(st) -> str:
    f = lambda e: (f"mean {e['mean']:.3f} p99.9 {e['p999']:.2f} max {e['max']:.2f}; near the row maximum mean {e['top_mean']:.3f} "
                   f"p99.9 {e['top_p999']:.2f} max {e['top_max']:.2f}")
    return (f"{st['rows']} rows x V logits, std {st['logit_std']:.3f}, absmax {st['absmax']:.2f}; |deviation| in bf16 ulps of the row's "
            f"largest |logit|:\n     HIP vs bf16 oracle   {f(st['hip_ref'])}\n     HIP vs fp32 oracle   {f(st['hip_tru'])}\n"
            f"     bf16 vs fp32 oracle  {f(st['ref_tru'])}\n     argmax agreement: HIP / bf16 oracle {st['agree_hip_ref']:.4f}, "
            f"HIP / fp32 {st['agree_hip_tru']:.4f}, bf16 oracle / fp32 {st['agree_ref_tru']:.4f} ({st['top_count']} logits within 16 ulps "
            f"of their row's maximum)")


def test_c2_full_size_forward_logits_vs_cpu_oracle(full):
    """CoDAModel.forward + lm_head (modeling_coda.py:312-355,406-412) at 28 layers, V = 151 936, logit for logit against the
    CPU oracle, on three states of the C2 run (all generation positions masked / half revealed / none masked), with the
    bench's lm_head x 8 and with the unscaled lm_head (|logit| < 8, a trained checkpoint's range).

    Two bf16 implementations of a 28-layer forward land on different bf16 values (every rounding point sits on another fp32
    sum), so the yardstick is the THIRD run: the same forward in fp32.  Asserted: (i) the HIP logits are as close to the
    fp32 logits as the reference's own bf16 arithmetic is (means within 10 %, p99.9 within 15 %), i.e. the deviation is
    the reference's bf16 noise floor and not an error of the port; (ii) the HIP / bf16-oracle deviation of the logits that
    can take part in a token decision stays within tests/parity.py's FWD_ERR_ULPS_FULL at p99.9 - the figure every parity
    band is derived from."""
    from parity import FWD_ERR_ULPS_FULL
    cfg, m, ids, seq, hist, trace = full
    d = ref_cpu.Dims()
    x = torch.stack([F.pad(ids[3], (0, G), value=cfg.mask_token_id), hist[63][11], seq[29]])
    for gain in (8.0, 1.0):
        with lm_head_gain(m, gain):
            sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
            st = forward_error_stats(m, sd, d, x)
        print(f"full-size forward, lm_head x {gain:g}: {describe_forward(st)}")
        hr, ht, rt = st["hip_ref"], st["hip_tru"], st["ref_tru"]
        assert ht["mean"] <= 1.10 * rt["mean"] and ht["top_mean"] <= 1.10 * rt["top_mean"], st
        assert ht["p999"] <= 1.15 * rt["p999"] and ht["row_rms_max"] <= 1.15 * rt["row_rms_max"], st
        assert hr["top_p999"] <= FWD_ERR_ULPS_FULL and hr["p999"] <= 1.25 * FWD_ERR_ULPS_FULL, st
        assert st["agree_hip_tru"] >= st["agree_ref_tru"] - 0.02, st


@pytest.mark.parametrize("gain", [8.0, 1.0])
def test_top_p_kept_set_on_the_logits_the_kernel_consumed(full, gain):
    """The top-p kernel pinned directly: bf16 logits of real candidate rows at V = 151 936 - exactly what
    `coda_confidence_from_logits` consumes on the C4 / lm-eval path - peaked (lm_head x 8) and flat (x 1: the nucleus
    boundary lies deep in a tail of near-equal logits).  maskgit_plus at T = 0 returns 1 / Z_kept, so the kept MASS is read
    back from the confidence to fp32 resolution (one flat-row token is 7e-6 of it) and compared with
      (i)  the reference rule (generation_utils.py:39-50) evaluated in float64 on the same logits: equal, up to ONE
           boundary token on the few rows whose running mass passes top_p within the kernel's 2^-40 fixed-point step;
      (ii) ref_cpu.top_p_filter itself (torch fp32 sort / softmax / cumsum) on rows whose boundary is away from a tie:
           the kept sets must be identical there (same kept mass to 2e-6).
    The entropy over the kept set is compared as a second fingerprint of the set."""
    cfg, m, ids, seq, hist, trace = full
    eng = m.hip_engine()
    top_p = 0.9
    lo, hi = P + BL, P + 2 * BL
    with lm_head_gain(m, gain):
        x = hist[20].to(DEV)                                                # block 1, 4 steps in
        eng.set_mask_id(cfg.mask_token_id)
        eng.forward_hidden(x, need=(lo - 1, hi - 1))
        eng.build_candidates(x, lo, hi)
        n = int(eng.count.item())
        rows = eng.candidate_logits(B * BL)[:n][::7][:96].contiguous().clone()      # bf16 [96, V]
        N = rows.shape[0]
        out = {}
        for alg in (0, 2):
            conf = torch.full((N,), float("nan"), device=DEV)
            x0 = torch.full((N,), -1, device=DEV, dtype=torch.int64)
            eng.confidence_from_logits(rows, alg, 0.0, top_p, None, None, 0, False, use_count=False, scatter=False, conf=conf, x0=x0)
            torch.cuda.synchronize()
            out[alg] = (conf.cpu().double(), x0.cpu())
    lf = rows.cpu().double()
    srt, idx = torch.sort(lf, dim=-1, descending=True, stable=True)
    e = torch.exp(srt - srt[:, :1])
    ps = e / e.sum(-1, keepdim=True)
    excl = torch.cumsum(ps, -1) - ps
    kept_sorted = excl <= top_p
    kept_sorted[:, 0] = True
    n_keep = kept_sorted.sum(-1)
    z_exact = (e * kept_sorted).sum(-1)
    e_boundary = e.gather(1, (n_keep - 1).clamp_min(0)[:, None]).squeeze(1)     # mass of the last kept token (rel. to the max)
    z_hip = 1.0 / out[0][0]
    assert torch.equal(out[0][1], lf.argmax(-1))
    d_exact = (z_hip - z_exact).abs()
    same = d_exact <= 2e-6 * z_exact
    one_off = d_exact <= 1.05 * e_boundary + 2e-6 * z_exact
    # (ii) the reference implementation on the same values (fp32): rows whose boundary is not within its cumsum's rounding
    ref_keep = ref_cpu.top_p_filter(rows.float().cpu(), top_p) > torch.finfo(torch.float32).min
    z_ref = (torch.exp(lf - srt[:, :1]) * ref_keep).sum(-1)
    margin = torch.minimum((excl.gather(1, (n_keep - 1)[:, None]).squeeze(1) - top_p).abs(),
                           (excl.gather(1, n_keep.clamp_max(lf.shape[1] - 1)[:, None]).squeeze(1) - top_p).abs())
    clear = margin > 2e-4
    same_ref = (z_hip - z_ref).abs() <= 2e-6 * z_ref
    # entropy over the kept set
    keep = torch.zeros_like(lf, dtype=torch.bool).scatter_(1, idx, kept_sorted)
    pr = torch.softmax(lf.masked_fill(~keep, -1e30), -1)
    ent = (pr * torch.log(pr + 1e-10)).sum(-1)
    d_ent = (out[2][0] - ent).abs()
    print(f"top-p 0.9 on {N} consumed rows, lm_head x {gain:g}: tokens kept per row {int(n_keep.min())}..{int(n_keep.max())}; kept mass equal "
          f"to the float64 rule on {int(same.sum())} rows, within one boundary token on {int(one_off.sum())}; {int(clear.sum())} rows "
          f"with a clear boundary, kept mass equal to ref_cpu.top_p_filter's on {int((same_ref & clear).sum())} of them; "
          f"max |entropy - float64 rule| {float(d_ent.max()):.2e}")
    assert one_off.all() and same.float().mean() >= 0.9
    assert (same_ref | ~clear).all()
    assert float(d_ent.max()) < 5e-4
