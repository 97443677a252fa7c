"""End-to-end parity of model.diffusion_generate (HIP path) against the CPU oracle and the
reference-generated golden fixtures.

Levels:
  L2 forward     bf16 logits vs the oracle's bf16 forward: stated tolerance below.
  L2 sampler     per-step TEACHER-FORCED comparison: the oracle is restarted from the HIP path's
                 state at every step, so one near-tie cannot cascade; a step may differ only where
                 the oracle's own decision margin is inside the declared band.
  free-running   final sequences vs the golden fixtures produced by the reference itself.
"""
import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
EXCUSED_SMALL = 0.08    # bound on the share of tokens the near-tie band may excuse on the 2- / 3-layer models (measured <= 2.5 %; T = 0.1 + top-p: 1.3 %)

from coda_amd import CoDAConfig, CoDALanguageModel, DLMGenerationConfig  # noqa: E402
from oracle import ref_cpu  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
SMALL = dict(vocab=1024, hidden=256, inter=512, heads=2, kv_heads=1, head_dim=128, layers=2, mask_id=1023, pad_id=1000,
             max_pos=4096)
MID = dict(vocab=8192, hidden=512, inter=1024, heads=4, kv_heads=2, head_dim=128, layers=3, mask_id=8191, pad_id=8000,
           max_pos=4096)


def build(dims_kw, seed, gain=8.0, jitter=0.1):
    d = ref_cpu.Dims(**dims_kw)
    sd32 = ref_cpu.make_state_dict(d, seed, torch.float32, lm_head_gain=gain, norm_jitter=jitter)
    sd = {k: v.to(BF) for k, v in sd32.items()}
    cfg = CoDAConfig(vocab_size=d.vocab, hidden_size=d.hidden, intermediate_size=d.inter, num_attention_heads=d.heads,
                     num_key_value_heads=d.kv_heads, head_dim=d.head_dim, num_hidden_layers=d.layers,
                     mask_token_id=d.mask_id, pad_token_id=d.pad_id, bos_token_id=d.pad_id, eos_token_id=d.pad_id + 1,
                     max_position_embeddings=d.max_pos)
    old = torch.get_default_dtype()
    torch.set_default_dtype(BF)
    try:
        m = CoDALanguageModel(cfg)
    finally:
        torch.set_default_dtype(old)
    m.load_state_dict(sd, strict=True)
    return d, sd, m.eval().to(DEV)


@pytest.fixture(scope="module")
def small():
    return build(SMALL, 22)


@pytest.fixture(scope="module")
def mid():
    return build(MID, 31)


def test_forward_logits_vs_oracle(small, mid):
    for d, sd, m in (small, mid):
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(0, d.pad_id, (3, 70), generator=g)
        ids[:, 40:] = d.mask_id
        with torch.no_grad():
            ref = ref_cpu.forward_logits(sd, d, ids).float()
            out, none = m(ids.to(DEV))
        assert none is None and out.dtype == BF and out.shape == ref.shape
        # attention_mask is accepted and ignored
        out2, _ = m(ids.to(DEV), attention_mask=(ids != d.pad_id).to(DEV))
        assert torch.equal(out, out2)
        err = (out.float().cpu() - ref).abs()
        scale = ref.std().item()
        print(f"forward L2: logits std {scale:.3f} max err {err.max():.4f} mean err {err.mean():.5f}")
        # bf16 tolerance: accumulation-order differences through the layers; stated relative to the logit scale
        # observed on MI355X: mean 0.4 %, max 2.6 % of the logit std; asserted at 3x that
        assert err.mean().item() < 0.012 * scale
        assert err.max().item() < 0.08 * scale
        agree = (out.float().cpu().argmax(-1) == ref.argmax(-1)).float().mean().item()
        assert agree > 0.97, agree
        # the same deviation in the unit tests/parity.py states its bands in: bf16 ulps of the row's largest |logit|
        from parity import FWD_ERR_ULPS_SMALL, bf16_ulp
        ulp = bf16_ulp(ref.abs().amax(-1, keepdim=True))
        eu = err / ulp
        top = eu[ref >= ref.amax(-1, keepdim=True) - 16 * ulp]
        kth = lambda t, q: float(torch.kthvalue(t.reshape(-1), max(1, int(q * t.numel())))[0])
        print(f"   in bf16 ulps of the row's largest |logit| ({d.layers} layers): mean {eu.mean():.3f} p99.9 {kth(eu, 0.999):.2f} max {eu.max():.2f}; "
              f"near the row maximum ({top.numel()} logits) p99.9 {kth(top, 0.999):.2f} max {top.max():.2f}")
        assert kth(top, 0.999) <= FWD_ERR_ULPS_SMALL and kth(eu, 0.999) <= 1.25 * FWD_ERR_ULPS_SMALL


@pytest.mark.parametrize("alg", ["entropy", "maskgit_plus", "topk_margin"])
@pytest.mark.parametrize("block_length", [None, 8])
def test_greedy_generate_teacher_forced(small, alg, block_length):
    from parity import assert_parity, describe, teacher_forced
    d, sd, m = small
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, d.pad_id, (4, 24), generator=g)
    kw = dict(max_new_tokens=32, steps=16, alg=alg)
    trace = []
    out = m.diffusion_generate(ids.to(DEV), generation_config=DLMGenerationConfig(
        mask_token_id=d.mask_id, pad_token_id=d.pad_id, return_dict_in_generate=True, output_history=True,
        block_length=block_length, **kw), debug_trace=trace)
    hist = [h.cpu() for h in out.history]
    assert len(hist) == 16 and out.sequences.shape == (4, 56)
    assert (out.sequences[:, :24].cpu() == ids).all()
    assert (out.sequences != d.mask_id).all()
    stats, tr = teacher_forced(sd, d, ids, hist, trace, block_length=block_length, **kw)
    print(f"teacher-forced {alg} bl={block_length}: {describe(stats)}")
    assert_parity(stats, max_excused=EXCUSED_SMALL)


def test_golden_free_running(small, golden):
    """Final sequences of the reference (bf16, CPU) vs the HIP path, free running, greedy."""
    d, sd, m = small
    keys = sorted(k[:-5] for k in golden.files if k.startswith("e2e_small_bf16") and k.endswith("_meta"))
    rates = []
    for key in keys:
        meta = ast.literal_eval(str(golden[key + "_meta"]))
        assert meta["weight_seed"] == 22 and meta["gain"] == 8.0
        ids = torch.from_numpy(golden[key + "_ids"])
        out = m.diffusion_generate(ids.to(DEV), max_new_tokens=meta["G"], steps=meta["steps"], alg=meta["alg"],
                                   mask_token_id=d.mask_id, round_conf_to="bf16")
        ref = torch.from_numpy(golden[key + "_seq"])
        rate = (out.cpu() == ref)[:, meta["P"]:].float().mean().item()
        rates.append(rate)
        # first step whose state differs from the reference's, and the share of identical step states
        hout = m.diffusion_generate(ids.to(DEV), max_new_tokens=meta["G"], steps=meta["steps"], alg=meta["alg"],
                                    mask_token_id=d.mask_id, round_conf_to="bf16", return_dict_in_generate=True, output_history=True)
        ref_hist = torch.from_numpy(golden[key + "_hist"]).long()
        same = [bool(torch.equal(h.cpu(), r)) for h, r in zip(hout.history, ref_hist)]
        first = same.index(False) if False in same else None
        print(f"golden {key} alg={meta['alg']} steps={meta['steps']}: token match {rate:.3f}; first diverging step "
              f"{first if first is not None else 'none'} of {len(same)}, {sum(same)}/{len(same)} step states identical")
        assert out.shape == ref.shape and (out.cpu()[:, :meta["P"]] == ref[:, :meta["P"]]).all()
    # free-running bf16 vs bf16-on-another-device is tie-limited (F5); report, and require broad agreement
    assert np.mean(rates) > 0.7, rates


def test_f9_argmax_is_mask_token_and_k_beyond_a_rows_candidates(small):
    """SURVEY.md F9 (generation_utils.py:507-512,528-538), through the logits hook with canned logits:
      (a) a position whose argmax IS the mask token is 'revealed' as mask, i.e. stays masked - same on both paths;
      (b) k comes from the batch-mean mask count, so rows couple: k equals the reference's fp32 value at every step;
      (c) DECLARED DEVIATION: when k exceeds a row's candidates the reference's topk runs into the -inf entries and
          writes mask_id over that many already-revealed tokens (which ones is torch.topk's unspecified order among
          equal keys); the HIP path reveals the row's candidates and touches nothing else.
    Teacher-forced per step from the HIP path's states; the canned logits are peaked, so no decision is a near-tie."""
    d, sd, m = small
    B, P, G, steps = 2, 4, 12, 6
    L, V = P + G, d.vocab
    g = torch.Generator().manual_seed(91)
    table = torch.randn(B, L, V, generator=g) * 6.0
    table[:, :, d.mask_id] = -30.0
    stuck = [5, 7, 8, 11]                                   # row 0: argmax = mask token, and the most confident rows
    for p in stuck:
        table[0, p] = -20.0
        table[0, p, d.mask_id] = 20.0
    ids = torch.randint(0, d.pad_id, (B, P), generator=g)

    def hook(step, x, logits):
        return table.to(logits.device)
    kw = dict(max_new_tokens=G, steps=steps, alg="entropy")
    trace = []
    out = m.diffusion_generate(ids.to(DEV), mask_token_id=d.mask_id, generation_logits_hook_func=hook,
                               return_dict_in_generate=True, output_history=True, debug_trace=trace, **kw)
    hist = [h.cpu() for h in out.history]
    x_init = torch.nn.functional.pad(ids, (0, G), value=d.mask_id)
    states = [x_init] + hist[:-1]
    args = ref_cpu.SamplerArgs(max_length=L, steps=steps, mask_id=d.mask_id, alg="entropy")
    tr = ref_cpu.sample_loop(lambda x: table, ids, args, logits_hook=lambda i, x, lg: table, trace=True, force_states=states)
    remasked_total = 0
    for i in range(steps):
        prev, hip, ora = states[i], hist[i], tr.history[i]
        was_mask = prev == d.mask_id
        assert int(trace[i]["k"].item()) == tr.k[i], (i, int(trace[i]["k"].item()), tr.k[i])               # (b)
        assert torch.equal(hip[was_mask], ora[was_mask]), i                                               # reveals identical
        assert (hip[0, stuck] == d.mask_id).all()                                                           # (a)
        assert torch.equal(hip[~was_mask], prev[~was_mask]), i                                              # HIP never touches a revealed token
        changed = (~was_mask) & (ora != prev)
        assert (ora[changed] == d.mask_id).all()
        for b in range(B):
            excess = max(0, tr.k[i] - int(was_mask[b].sum()))
            assert int(changed[b].sum()) == excess, (i, b, int(changed[b].sum()), excess)                  # (c)
        remasked_total += int(changed.sum())
    print(f"F9 case: k per step {tr.k}; the reference re-masked {remasked_total} revealed tokens over the run (declared deviation), "
          f"stuck positions still masked at the end: {int((hist[-1][0, stuck] == d.mask_id).sum())}/4")
    assert remasked_total > 0, "the case never drove k beyond a row's candidates"
    assert (hist[-1][1] != d.mask_id).all()


def test_separated_golden_free_running_token_exact(small, mid, separated):
    """Greedy runs of the REFERENCE (bf16, CPU) that are well posed - no decision on a near-tie: no step's k-th and
    (k+1)-th confidences are equal and the whole run survives 16 reruns with forward-sized noise on the logits
    (make_golden.py section G9; the selection uses CPU-side stability only, nothing from the HIP path).
    The HIP path, free running with its default settings, must give the reference's tokens at every step, exactly.
    Windowed cases (block_length; no reference implementation) hold the oracle's outputs, same selection rule."""
    keys = sorted(k[:-5] for k in separated.files if k.endswith("_meta"))
    assert len(keys) >= 6
    for key in keys:
        meta = ast.literal_eval(str(separated[key + "_meta"]))
        # "mid" vectors run at T = B * L >= 256 tokens: persistent GEMM engines + fused QKV / q-k-norm / RoPE epilogue
        d, sd, m = small if meta["dims"] == SMALL else mid
        assert meta["gain"] == 8.0 and meta["dims"] in (SMALL, MID) and meta["weight_seed"] == (22 if meta["dims"] == SMALL else 31)
        ids = torch.from_numpy(separated[key + "_ids"])
        out = m.diffusion_generate(ids.to(DEV), max_new_tokens=meta["G"], steps=meta["steps"], alg=meta["alg"],
                                   block_length=meta.get("block_length"), mask_token_id=d.mask_id,
                                   return_dict_in_generate=True, output_history=True)
        ref_seq = torch.from_numpy(separated[key + "_seq"])
        ref_hist = torch.from_numpy(separated[key + "_hist"]).long()
        hist = torch.stack([h.cpu() for h in out.history])
        steps_equal = int((hist == ref_hist).all(dim=(1, 2)).sum())
        print(f"separated {key} alg={meta['alg']} B={meta['B']} G={meta['G']} steps={meta['steps']} block={meta.get('block_length')}: "
              f"{steps_equal}/{meta['steps']} step states identical, final tokens equal: {torch.equal(out.sequences.cpu(), ref_seq)}")
        assert torch.equal(out.sequences.cpu(), ref_seq), key
        assert torch.equal(hist, ref_hist), key


@pytest.mark.parametrize("block_length", [None, 8])
def test_last_layer_compact_tail_same_tokens(small, mid, block_length):
    """engine.forward_hidden(need=...): last layer's o_proj / MLP / norms on the consumed rows only == full forward."""
    for d, sd, m in (small, mid):
        g = torch.Generator().manual_seed(9)
        ids = torch.randint(0, d.pad_id, (5, 40), generator=g).to(DEV)
        kw = dict(max_new_tokens=32, steps=16, alg="entropy", mask_token_id=d.mask_id, block_length=block_length,
                  return_dict_in_generate=True, output_history=True)
        eng = m.hip_engine()
        a = m.diffusion_generate(ids, **kw)
        eng.prune_last_layer = False
        try:
            b = m.diffusion_generate(ids, **kw)
        finally:
            eng.prune_last_layer = True
        assert torch.equal(a.sequences, b.sequences)
        assert all(torch.equal(u, v) for u, v in zip(a.history, b.history))


@pytest.mark.parametrize("temperature", [0.0, 0.8])
def test_lmhead_routes_agree(small, mid, temperature):
    """engine.lmhead_confidence has two routes to the same numbers: the fused kernel (few candidate rows) and, from
    engine.materialise_min_rows on, row gather + counted GEMM + confidence kernel over a chunked bf16 logits buffer (faster
    for long candidate lists).  Forced both ways, with chunks so small that a step needs several and the last is ragged:
    same tokens, same histories - greedy and with the in-kernel noise (same hash keys on both routes)."""
    for d, sd, m in (small, mid):
        eng = m.hip_engine()
        g = torch.Generator().manual_seed(29)
        ids = torch.randint(0, d.pad_id, (5, 20), generator=g).to(DEV)
        kw = dict(max_new_tokens=24, steps=12, alg="topk_margin", alg_temp=0.2, temperature=temperature,
                  mask_token_id=d.mask_id, return_dict_in_generate=True, output_history=True)
        keep = (eng.materialise_min_rows, eng.logits_chunk_rows)
        outs = []
        try:
            for min_rows, chunk in ((1 << 30, 8192), (1, 8192), (1, 56)):
                eng.materialise_min_rows, eng.logits_chunk_rows = min_rows, chunk
                torch.manual_seed(1234)                      # the run's noise seed is drawn from torch's CPU generator
                outs.append(m.diffusion_generate(ids, **kw))
        finally:
            eng.materialise_min_rows, eng.logits_chunk_rows = keep
        for o in outs[1:]:
            assert torch.equal(o.sequences, outs[0].sequences)
            assert all(torch.equal(u, v) for u, v in zip(o.history, outs[0].history))
    if temperature > 0:
        # injected token noise (one [candidates, V] tensor per step): every chunk must read its own rows of it
        d, sd, m = small
        eng = m.hip_engine()
        g = torch.Generator().manual_seed(7)
        ids = torch.randint(0, d.pad_id, (3, 8), generator=g)
        kw = dict(max_new_tokens=16, steps=8, alg="maskgit_plus", temperature=temperature)
        torch.manual_seed(123)
        noise = ref_cpu.TorchNoise()
        ref_cpu.diffusion_generate(sd, d, ids, noise=noise, **kw)
        keep = (eng.materialise_min_rows, eng.logits_chunk_rows)
        outs = []
        try:
            for min_rows, chunk in ((1 << 30, 8192), (1, 20)):
                eng.materialise_min_rows, eng.logits_chunk_rows = min_rows, chunk
                outs.append(m.diffusion_generate(ids.to(DEV), mask_token_id=d.mask_id, noise_source=Replay(noise.log, d.vocab),
                                                 return_dict_in_generate=True, output_history=True, **kw))
        finally:
            eng.materialise_min_rows, eng.logits_chunk_rows = keep
        assert torch.equal(outs[0].sequences, outs[1].sequences)
        assert all(torch.equal(u, v) for u, v in zip(outs[0].history, outs[1].history))


def test_fused_equals_materialised_path(small):
    d, sd, m = small
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, d.pad_id, (3, 16), generator=g).to(DEV)
    kw = dict(max_new_tokens=32, steps=8, alg="entropy", mask_token_id=d.mask_id)
    a = m.diffusion_generate(ids, **kw)
    seen = []

    def hook(step, x, logits):
        seen.append((step, tuple(logits.shape), logits.dtype))
        return logits
    b = m.diffusion_generate(ids, generation_logits_hook_func=hook, **kw)
    assert len(seen) == 8 and seen[0][1] == (3, 48, d.vocab)
    assert (a != b).sum().item() <= 1
    c = m.diffusion_generate(ids, top_k=d.vocab, **kw)             # top-k over everything: unfused row path
    assert (a != c).sum().item() <= 1


def test_hooks_and_history_contract(small):
    d, sd, m = small
    ids = torch.randint(0, 900, (2, 8)).to(DEV)
    calls = []

    def tok_hook(step, x, logits):
        calls.append(step)
        assert (logits is None) == (step is None)
        return x
    out = m.diffusion_generate(ids, max_new_tokens=16, steps=4, alg="maskgit_plus", mask_token_id=d.mask_id,
                               generation_tokens_hook_func=tok_hook, return_dict_in_generate=True, output_history=True)
    assert calls == [None, 0, 1, 2, 3] and len(out.history) == 4
    assert ids.shape == (2, 8)                                     # caller's tensor untouched
    out2 = m.diffusion_generate(ids, max_new_tokens=16, steps=4, alg="maskgit_plus", mask_token_id=d.mask_id,
                                num_return_sequences=2)
    assert out2.shape == (4, 24)
    with pytest.raises(RuntimeError, match="Unknown alg"):
        m.diffusion_generate(ids, max_new_tokens=16, steps=4, alg="nope", mask_token_id=d.mask_id)
    with pytest.raises(ValueError):
        m.diffusion_generate(ids, max_length=8, steps=4, alg="entropy", mask_token_id=d.mask_id)


def test_hooks_fast_paths(small):
    """SURVEY 8(f)4 / generation_utils.py:453,460,541.  (i) a TOKENS hook alone keeps the fused, logits-free route: it is
    handed a LazyLogits that stays unmaterialised unless read, and the run equals the hook-free run; (ii) a tokens hook that
    does read it gets the step's shifted logits - the same tensor a logits hook sees; (iii) the logits hook's tensor is
    `cat(logits[:, :1], logits[:, :-1])` of model(x) (built without that copy) and lives in one engine buffer."""
    from coda_amd.generation_utils import LazyLogits
    d, sd, m = small
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, d.pad_id, (3, 16), generator=g).to(DEV)
    kw = dict(max_new_tokens=32, steps=8, alg="entropy", mask_token_id=d.mask_id, block_length=8)
    plain = m.diffusion_generate(ids, **kw)
    lazies = []

    def tok_hook(step, x, logits):
        if step is not None:
            lazies.append(logits)
        return x
    hooked = m.diffusion_generate(ids, generation_tokens_hook_func=tok_hook, **kw)
    assert torch.equal(plain, hooked)
    assert len(lazies) == 8 and all(isinstance(z, LazyLogits) and not z.materialized for z in lazies)

    seen_tok, seen_log, states = [], [], []

    def reading_hook(step, x, logits):
        if step is not None:
            assert tuple(logits.shape) == (3, 48, d.vocab) and logits.dtype == BF        # attribute access materialises
            seen_tok.append((logits[:, 5].clone(), torch.softmax(logits.float(), -1).amax(-1), (logits * 2)[0, 0, :4].clone()))
        return x

    def log_hook(step, x, logits):
        states.append(x.clone())
        seen_log.append(logits.clone())
        return logits
    a = m.diffusion_generate(ids, generation_tokens_hook_func=reading_hook, **kw)
    b = m.diffusion_generate(ids, generation_logits_hook_func=log_hook, **kw)
    assert torch.equal(a, plain)
    assert (b != plain).sum().item() <= 1          # materialised vs fused route: same logits, confidences to 1e-6 (near-ties)
    if torch.equal(b, plain):
        for (col, pmax, twice), full in zip(seen_tok, seen_log):
            assert torch.equal(col, full[:, 5]) and torch.equal(twice, (full * 2)[0, 0, :4])
            assert torch.equal(pmax, torch.softmax(full.float(), -1).amax(-1))
    for x, full in zip(states, seen_log):
        ref, _ = m(x)
        assert torch.equal(full, torch.cat([ref[:, :1], ref[:, :-1]], dim=1))
    # a hook that rewrites the logits steers the sampler: force token 7 everywhere
    def force7(step, x, logits):
        out = torch.full_like(logits, -30.0)
        out[..., 7] = 30.0
        return out
    c = m.diffusion_generate(ids, generation_logits_hook_func=force7, **kw)
    assert (c[:, 16:] == 7).all() and torch.equal(c[:, :16], ids)
    # ... also when it returns another dtype (fp32): compacted by torch, same result
    c32 = m.diffusion_generate(ids, generation_logits_hook_func=lambda s_, x, l: force7(s_, x, l).float(), **kw)
    assert torch.equal(c32, c)
    with pytest.raises(Exception):
        m.diffusion_generate(ids, generation_tokens_hook_func=lambda s_, x, l: x.cpu(), **kw)


class Replay:
    """Feeds noise recorded from an oracle run (CPU tensors) to the HIP path."""

    def __init__(self, log, V):
        self.by = {(k, s): t for k, s, t in log}
        self.V = V

    def token_noise(self, step, cap, V, picked=None):
        q = self.by.get(("token", step))
        out = torch.ones(cap, V)
        if q is not None:
            if picked is None:
                out[: q.shape[0]] = q.float()
            else:
                idx = picked.cpu().nonzero().squeeze(1)[: q.shape[0]]
                out[idx] = q.float()
        return out.to(DEV)

    def step_seed(self, step):
        return 0

    def select_noise(self, step, B, L):
        return self.by[("select", step)].float().to(DEV).contiguous()

    def origin_uniform(self, step, cap):
        u = self.by[("origin", step)]
        out = torch.ones(cap)
        out[: u.shape[0]] = u
        return out.to(DEV)


@pytest.mark.parametrize("cfg", [
    dict(alg="entropy", alg_temp=0.1),
    dict(alg="maskgit_plus", temperature=0.7),
    dict(alg="maskgit_plus", temperature=0.7, top_p=0.9),
    dict(alg="topk_margin", temperature=0.5, top_k=20, alg_temp=0.2),
    dict(alg="origin"),
    dict(alg="origin", temperature=0.8),
])
def test_stochastic_with_injected_noise(small, cfg):
    """Noise injection: the oracle draws Exp(1)/uniform noise from a seeded CPU generator; the same
    tensors are replayed into the HIP kernels.  Outputs must then match like the greedy case."""
    d, sd, m = small
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, d.pad_id, (2, 8), generator=g)
    kw = dict(max_new_tokens=16, steps=8, **cfg)
    torch.manual_seed(123)
    noise = ref_cpu.TorchNoise()
    ref = ref_cpu.diffusion_generate(sd, d, ids, noise=noise, **kw)
    trace = []
    out = m.diffusion_generate(ids.to(DEV), mask_token_id=d.mask_id, noise_source=Replay(noise.log, d.vocab),
                               return_dict_in_generate=True, output_history=True, debug_trace=trace, **kw)
    from parity import assert_parity, describe, teacher_forced
    hist = [h.cpu() for h in out.history]
    stats, tr = teacher_forced(sd, d, ids, hist, trace, noise_log=noise.log, **kw)
    rate = (out.sequences.cpu() == ref.sequences).float().mean().item()
    print(f"stochastic {cfg}: free-running token match {rate:.3f}; teacher-forced {describe(stats)}")
    assert_parity(stats, max_excused=EXCUSED_SMALL)


def test_device_rng_is_reproducible(small):
    d, sd, m = small
    ids = torch.randint(0, 900, (2, 8)).to(DEV)
    kw = dict(max_new_tokens=16, steps=8, alg="entropy", alg_temp=0.2, temperature=0.9, mask_token_id=d.mask_id)
    torch.manual_seed(5)
    a = m.diffusion_generate(ids, **kw)
    torch.manual_seed(5)
    b = m.diffusion_generate(ids, **kw)
    torch.manual_seed(6)
    c = m.diffusion_generate(ids, **kw)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert (a != d.mask_id).all()


@pytest.mark.parametrize("temperature", [0.0, 0.1])
def test_lm_eval_adapter_left_padded(small, temperature):
    """the lm-eval adapter's call shape (diffllm.py:265-295) on the HIP path - unequal prompts left-padded
    and cut to max_prompt_len, attention_mask passed, top_p = 0.9 with T = 0 (deterministic) and the eval scripts'
    T = 0.1 (noise replayed from the oracle)."""
    from coda_amd.lm_eval_adapter import DiffLLMAdapter
    from parity import assert_parity, describe, teacher_forced
    d, sd, m = small
    g = torch.Generator().manual_seed(77)
    prompts = [torch.randint(0, d.pad_id, (n,), generator=g).tolist() for n in (9, 4, 12, 1)]
    a = DiffLLMAdapter(m, batch_size=4, max_prompt_len=10, max_new_tokens=16, diffusion_steps=16, temperature=temperature, top_p=0.9)
    ids, attn = a.prepare_batch(prompts)
    assert ids.shape == (4, 10) and (ids[3, :9] == d.pad_id).all() and attn.dtype == torch.bool
    kw = dict(max_new_tokens=16, steps=16, alg="entropy", temperature=temperature, top_p=0.9, alg_temp=0.0)
    extra, log = {}, None
    if temperature > 0:
        torch.manual_seed(321)
        noise = ref_cpu.TorchNoise()
        ref_cpu.diffusion_generate(sd, d, ids.cpu(), noise=noise, **kw)
        extra, log = dict(noise_source=Replay(noise.log, d.vocab)), noise.log
    gk = a.generation_kwargs()
    gk.update(output_history=True)
    trace = []
    out = m.diffusion_generate(ids, attention_mask=attn, debug_trace=trace, **extra, **gk)
    assert (out.sequences[:, :10] == ids).all() and (out.sequences != d.mask_id).all()
    stats, _ = teacher_forced(sd, d, ids.cpu(), [h.cpu() for h in out.history], trace, noise_log=log, **kw)
    print(f"lm-eval shape T={temperature}: {describe(stats)}")
    assert_parity(stats, max_excused=EXCUSED_SMALL)
    if temperature == 0:
        resp = a.generate_batch_ids(prompts)
        eos = d.pad_id + 1
        for r, row in zip(resp, out.sequences.cpu()):
            tail = row[10:].tolist()
            assert r == (tail[:tail.index(eos)] if eos in tail else tail)


@pytest.mark.parametrize("cfg", [
    dict(steps=8, temperature=0.0, threshold=0.9),
    dict(steps=4, temperature=0.0, threshold=0.5),
    dict(steps=16, temperature=0.0, threshold=0.02),
    dict(steps=8, temperature=0.6, top_p=0.95, threshold=0.7),
])
def test_confidence_threshold_serving_alg(small, cfg):
    """the serving sampler's confidence_threshold alg (serving/dream/model/generation_utils.py:416-465),
    one sequence, data-dependent step count; teacher-forced against the oracle, bookkeeping replayed on HIP confidences."""
    from coda_amd.serving import SamplerService
    from parity import assert_parity, describe, teacher_forced
    d, sd, m = small
    g = torch.Generator().manual_seed(52)
    ids = torch.randint(0, d.pad_id, (1, 7), generator=g)
    cfg = dict(cfg)
    kw = dict(max_new_tokens=16, alg="confidence_threshold", **cfg)
    extra, log = {}, None
    if cfg["temperature"] > 0:
        torch.manual_seed(99)
        noise = ref_cpu.TorchNoise()
        ref_cpu.diffusion_generate(sd, d, ids, noise=noise, **kw)
        extra, log = dict(noise_source=Replay(noise.log, d.vocab)), noise.log
    trace = []
    out = m.diffusion_generate(ids.to(DEV), mask_token_id=d.mask_id, return_dict_in_generate=True, output_history=True,
                               debug_trace=trace, **extra, **kw)
    hist = [h.cpu() for h in out.history]
    assert (out.sequences != d.mask_id).all() and len(hist) >= cfg["steps"] and len(trace) == len(hist)
    stats, _ = teacher_forced(sd, d, ids, hist, trace, noise_log=log, **kw)
    print(f"confidence_threshold {cfg}: {len(hist)} steps used; {describe(stats)}")
    assert_parity(stats, max_excused=EXCUSED_SMALL)
    if cfg["temperature"] == 0:
        # the serving mirror returns the same completion, mask/pad ids dropped
        svc = SamplerService(m, mask_token_id=d.mask_id)
        got = svc.generate_ids(ids[0].tolist(), max_tokens=16, steps=cfg["steps"], alg="confidence_threshold", alg_temp=0.0,
                               threshold=cfg["threshold"])
        tail = out.sequences[0, 7:].cpu()
        assert got == tail[(tail != d.mask_id) & (tail != d.pad_id)].tolist()
    with pytest.raises(AssertionError):
        m.diffusion_generate(ids.repeat(2, 1).to(DEV), mask_token_id=d.mask_id, **kw)           # batch size must be 1
    with pytest.raises(AssertionError):
        m.diffusion_generate(ids.to(DEV), mask_token_id=d.mask_id, max_new_tokens=16, steps=5, alg="confidence_threshold")


def test_load_packed_checkpoint_equals_module_path(small, tmp_path):
    """safetensors (sharded) -> fused HBM layouts directly; the nn.Module tree holds no storage and the
    generated tokens equal those of the same weights loaded through the ordinary parameter path."""
    import json
    import os
    from safetensors.torch import save_file
    from coda_amd import checkpoint
    d, sd, m = small
    raw = str(tmp_path)
    keys = list(sd)
    wm = {}
    for i, ks in enumerate((keys[0::2], keys[1::2])):
        fn = f"model-{i + 1:05d}-of-00002.safetensors"
        save_file({k: sd[k].contiguous() for k in ks}, os.path.join(raw, fn))
        wm.update({k: fn for k in ks})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(raw, checkpoint.INDEX_NAME), "w"))
    m.config.save_pretrained(raw)
    packed = checkpoint.load_packed(raw, device=DEV)
    assert all(p.device.type == "meta" for p in packed.parameters())
    assert packed.device.type == "cuda"
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, d.pad_id, (3, 12), generator=g).to(DEV)
    kw = dict(max_new_tokens=24, steps=12, alg="entropy", mask_token_id=d.mask_id, block_length=8)
    a = m.diffusion_generate(ids, **kw)
    b = packed.diffusion_generate(ids, **kw)
    assert torch.equal(a, b)
    la, _ = m(ids)
    lb, _ = packed(ids)
    assert torch.equal(la, lb)
