"""The CPU oracle against fixtures produced by the reference itself (tests/golden/make_golden.py).

Every array named *_seq / *_hist / *_logits / st_* in reference_vectors.npz is an OUTPUT OF THE
REFERENCE sampler; this test regenerates the weights from their seed and requires the oracle to
reproduce them: bit-exact tokens and histories, logits to 1e-5 (they were 0 when generated)."""
import ast

import numpy as np
import pytest
import torch

from oracle import ref_cpu

DT = {"torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16}


def _meta(golden, key):
    return ast.literal_eval(str(golden[key]))


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_forward_logits(golden, name):
    meta = _meta(golden, f"fwd_{name}_meta")
    d = ref_cpu.Dims(**meta["dims"])
    sd = ref_cpu.make_state_dict(d, meta["weight_seed"], torch.float32, meta["gain"], meta["norm_jitter"])
    ids = torch.from_numpy(golden[f"fwd_{name}_ids"])
    with torch.no_grad():
        mine = ref_cpu.forward_logits(sd, d, ids)
    ref = torch.from_numpy(golden[f"fwd_{name}_logits"])
    assert (mine - ref).abs().max().item() < 1e-5


def _e2e_keys(golden, prefix):
    return sorted(k[:-5] for k in golden.files if k.startswith(prefix) and k.endswith("_meta"))


@pytest.mark.parametrize("prefix", ["e2e_tiny_f32", "e2e_small_f32", "e2e_small_bf16"])
def test_end_to_end_token_exact(golden, prefix):
    keys = _e2e_keys(golden, prefix)
    assert keys
    sd_cache = {}
    for key in keys:
        meta = _meta(golden, key + "_meta")
        d = ref_cpu.Dims(**meta["dims"])
        dt = DT[meta["dtype"]]
        ck = (meta["weight_seed"], meta["dtype"])
        if ck not in sd_cache:
            sd32 = ref_cpu.make_state_dict(d, meta["weight_seed"], torch.float32, meta["gain"], meta["norm_jitter"])
            sd_cache[ck] = {k: v.to(dt) for k, v in sd32.items()}
        kw = {k: meta[k] for k in ("steps", "alg", "alg_temp", "temperature", "top_p", "top_k") if k in meta}
        if meta["seed"] is not None:
            torch.manual_seed(meta["seed"])
        tr = ref_cpu.diffusion_generate(sd_cache[ck], d, torch.from_numpy(golden[key + "_ids"]),
                                        max_new_tokens=meta["G"], **kw)
        assert np.array_equal(tr.sequences.numpy(), golden[key + "_seq"]), key
        assert np.array_equal(torch.stack(tr.history).numpy(), golden[key + "_hist"]), key


def test_block_length_equal_to_gen_length_is_reference(golden):
    key = "e2e_tiny_f32_0"
    meta = _meta(golden, key + "_meta")
    d = ref_cpu.Dims(**meta["dims"])
    sd = ref_cpu.make_state_dict(d, meta["weight_seed"], torch.float32, meta["gain"], meta["norm_jitter"])
    tr = ref_cpu.diffusion_generate(sd, d, torch.from_numpy(golden[key + "_ids"]), max_new_tokens=meta["G"],
                                    steps=meta["steps"], alg=meta["alg"], block_length=meta["G"])
    assert np.array_equal(tr.sequences.numpy(), golden[key + "_seq"])


def test_sample_tokens_flavours(golden):
    logits = torch.from_numpy(golden["st_logits"])
    cases = {"plain": {}, "margin": dict(margin=True), "entropy": dict(neg_entropy=True), "topp": dict(top_p=0.9),
             "topk": dict(top_k=50), "topp_topk_ent": dict(top_p=0.7, top_k=30, neg_entropy=True)}
    for nm, kw in cases.items():
        c, x0 = ref_cpu.sample_tokens(logits, 0.0, kw.pop("top_p", None), kw.pop("top_k", None), **kw)
        assert np.array_equal(x0.numpy(), golden[f"st_{nm}_x0"]), nm
        assert np.array_equal(c.numpy(), golden[f"st_{nm}_conf"]), nm
    assert np.array_equal(ref_cpu.top_p_filter(logits, 0.9).numpy(), golden["st_topp09"])
    assert np.array_equal(ref_cpu.top_k_filter(logits, 50).numpy(), golden["st_topk50"])


def test_top_p_known_answer(golden):
    """top_p_logits([2,1,.5,0,-1], 0.9) keeps 4, last -> finfo.min."""
    out = ref_cpu.top_p_filter(torch.tensor([[2.0, 1.0, 0.5, 0.0, -1.0]]), 0.9)
    assert np.array_equal(out.numpy(), golden["st_known_topp"])
    assert out[0, 4].item() == torch.finfo(torch.float32).min and (out[0, :4] == torch.tensor([2.0, 1.0, 0.5, 0.0])).all()
    p = torch.softmax(out, -1)[0]
    assert torch.allclose(p, torch.tensor([0.5793, 0.2131, 0.1293, 0.0784, 0.0]), atol=1e-4)
    conf, _ = ref_cpu.sample_tokens(torch.tensor([[2.0, 1.0, 0.5, 0.0, -1.0]]), 0.0, 0.9, None, neg_entropy=True)
    assert abs(conf.item() - (-1.10977)) < 1e-4


@pytest.mark.parametrize("G,steps,expect", [
    (256, 128, [1] + [2] * 126 + [3]),
    (512, 32, [15] + [16] * 30 + [17]),
    (1024, 512, [1, 1] + [2] * 509 + [4]),
    (32, 16, [1] + [2] * 14 + [3]),
    (64, 32, [1] + [2] * 30 + [3]),
    (768, 768, [0] + [1] * 766 + [2]),
])
def test_schedule_known_answers(G, steps, expect):
    """transfer counts computed with the reference's fp32 arithmetic."""
    assert ref_cpu.transfer_counts(G * 4, 4, steps, 1e-3) == expect


def test_stub_sampler_histories(golden):
    """Sampler-only vectors: the reference _sample driven by a canned-logits stub model."""
    V, mask_id = 2048, 2047
    g = torch.Generator().manual_seed(3)
    for scale in (0.05, 1.0, 6.0):
        table = torch.randn(V, V, generator=g) * scale
        table[:, mask_id] = -50.0

        def logits_fn(x, table=table):
            pos = torch.arange(x.shape[1])[None, :, None]
            return table[x] + 0.37 * torch.sin(pos * 0.91 + torch.arange(V)[None, None] * 0.13)

        for alg, extra in (("entropy", {}), ("maskgit_plus", {}), ("topk_margin", {}),
                           ("entropy", dict(alg_temp=0.3, temperature=0.6, top_p=0.8))):
            ids = torch.randint(0, 2000, (3, 6), generator=g)
            torch.manual_seed(99)
            args = ref_cpu.SamplerArgs(max_length=30, steps=6, mask_id=mask_id, alg=alg, output_history=True, **extra)
            tr = ref_cpu.sample_loop(logits_fn, ids, args)
            tag = f"stub_{scale}_{alg}_{len(extra)}"
            assert np.array_equal(ids.numpy(), golden[tag + "_ids"])
            assert np.array_equal(torch.stack(tr.history).numpy(), golden[tag + "_hist"]), tag


def test_online_forms_match_sample_tokens():
    """the single-pass forms the fused lm-head kernel carries, vs sample_tokens (1e-4 abs)."""
    g = torch.Generator().manual_seed(5)
    for scale in (0.05, 1.0, 4.0, 12.0):
        logits = torch.randn(16, 4096, generator=g) * scale
        for alg, kw in (("maskgit_plus", {}), ("topk_margin", dict(margin=True)), ("entropy", dict(neg_entropy=True))):
            c, x0 = ref_cpu.sample_tokens(logits, 0.0, **kw)
            oc, ox0 = ref_cpu.online_confidence(logits, alg)
            assert torch.equal(x0, ox0)
            assert (c.double() - oc).abs().max().item() < 1e-4, (scale, alg)
    q = torch.empty(16, 4096).exponential_(1, generator=g)
    logits = torch.randn(16, 4096, generator=g) * 3

    class N:
        def token_exp(self, step, probs):
            return q
    c, x0 = ref_cpu.sample_tokens(logits, 0.7, noise=N())
    oc, ox0 = ref_cpu.online_confidence(logits, "maskgit_plus", 0.7, q)
    assert torch.equal(x0, ox0) and (c.double() - oc).abs().max().item() < 1e-4


@pytest.mark.parametrize("prefix", ["lmeval_tiny_f32", "lmeval_small_f32", "lmeval_small_bf16"])
def test_lm_eval_call_shape(golden, prefix):
    """Left-padded, truncated prompts + attention_mask + the eval scripts' sampling settings
    (diffllm.py:265-295), as run through the reference model: tokens and per-step histories exact."""
    ids = torch.from_numpy(golden[prefix + "_ids"])
    plist = ast.literal_eval(str(golden[prefix + "_prompts"]))
    keys = _e2e_keys(golden, prefix)
    assert len(keys) == 3
    for key in keys:
        meta = _meta(golden, key + "_meta")
        d = ref_cpu.Dims(**meta["dims"])
        dt = DT[meta["dtype"]]
        sd = ref_cpu.make_state_dict(d, meta["weight_seed"], torch.float32, meta["gain"], meta["norm_jitter"])
        sd = {k: v.to(dt) for k, v in sd.items()}
        kw = {k: meta[k] for k in ("steps", "temperature", "top_p", "alg", "alg_temp") if k in meta}
        if "top_k" in meta:
            kw["top_k"] = meta["top_k"]

        def gen(pids, attn):
            assert torch.equal(pids, ids) and torch.equal(attn, ids.ne(d.pad_id))
            if meta["seed"] is not None:
                torch.manual_seed(meta["seed"])
            tr = ref_cpu.diffusion_generate(sd, d, pids, max_new_tokens=meta["G"], **kw)
            assert torch.equal(torch.stack(tr.history), torch.from_numpy(golden[key + "_hist"]).long())
            return tr.sequences

        # the id-level adapter restatement: padding/truncation reproduce the stored ids, the continuation is
        # the reference's sequence after the padded width, cut before the first eos
        resp = ref_cpu.generate_batch_ids(gen, plist, pad_id=d.pad_id, eos_id=d.pad_id + 1, max_prompt_len=meta["max_prompt_len"])
        ref_seq = torch.from_numpy(golden[key + "_seq"])
        for r, row in zip(resp, ref_seq):
            tail = row[ids.shape[1]:].tolist()
            assert r == (tail[:tail.index(d.pad_id + 1)] if d.pad_id + 1 in tail else tail)


@pytest.mark.parametrize("prefix", ["serve_tiny_f32", "serve_small_f32", "serve_small_bf16"])
def test_serving_sampler_vectors(golden, prefix):
    """The serving sampler (serving/dream/model/generation_utils.py:369-508) bound onto the reference CoDA model as
    service.py does: FastAPI defaults and confidence_threshold, including its data-dependent number of steps."""
    keys = _e2e_keys(golden, prefix)
    assert len(keys) == 5
    grew = 0
    for key in keys:
        meta = _meta(golden, key + "_meta")
        d = ref_cpu.Dims(**meta["dims"])
        sd = ref_cpu.make_state_dict(d, meta["weight_seed"], torch.float32, meta["gain"], meta["norm_jitter"])
        sd = {k: v.to(DT[meta["dtype"]]) for k, v in sd.items()}
        kw = {k: meta[k] for k in ("steps", "temperature", "alg", "alg_temp", "top_p") if k in meta}
        if meta["threshold"] is not None:
            kw["threshold"] = meta["threshold"]
        if meta["seed"] is not None:
            torch.manual_seed(meta["seed"])
        tr = ref_cpu.diffusion_generate(sd, d, torch.from_numpy(golden[key + "_ids"]), max_new_tokens=meta["G"], **kw)
        hist = torch.from_numpy(golden[key + "_hist"]).long()
        assert len(tr.history) == hist.shape[0]
        assert torch.equal(torch.stack(tr.history), hist)
        assert torch.equal(tr.sequences, torch.from_numpy(golden[key + "_seq"]))
        grew += hist.shape[0] > meta["steps"]
    assert grew >= 2          # the fixtures exercise the step-count extension


def test_teacher_forced_step_subset_equals_full_walk():
    """force_states + step_indices (the bounded full-size sample of tests/test_hip_fullsize.py) reproduces exactly
    the corresponding steps of a full teacher-forced walk, window mode included."""
    d = ref_cpu.Dims(vocab=512, hidden=256, inter=256, heads=2, kv_heads=1, head_dim=128, layers=1, mask_id=511, pad_id=500,
                     max_pos=512)
    sd = ref_cpu.make_state_dict(d, 9, torch.float32, lm_head_gain=8.0)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 500, (2, 8), generator=g)
    kw = dict(max_new_tokens=16, steps=8, alg="entropy", block_length=8)
    free = ref_cpu.diffusion_generate(sd, d, ids, trace=True, **kw)
    x_init = torch.nn.functional.pad(ids, (0, 16), value=d.mask_id)
    states = [x_init] + free.history[:-1]
    pick = [0, 3, 4, 7]
    sub = ref_cpu.diffusion_generate(sd, d, ids, trace=True, force_states=[states[i] for i in pick], step_indices=pick, **kw)
    assert len(sub.history) == len(pick)
    for n, i in enumerate(pick):
        assert torch.equal(sub.history[n], free.history[i])
        assert sub.k[n] == free.k[i] and torch.equal(sub.conf[n], free.conf[i]) and torch.equal(sub.cand[n], free.cand[i])


def test_separated_vectors_token_exact(separated):
    """Section G9 of make_golden.py: the oracle reproduces the reference's well-separated greedy runs."""
    keys = sorted(k[:-5] for k in separated.files if k.endswith("_meta"))
    assert len(keys) >= 6
    models = {}
    for key in keys:
        meta = _meta(separated, key + "_meta")
        mk = (repr(meta["dims"]), meta["weight_seed"])
        if mk not in models:
            d = ref_cpu.Dims(**meta["dims"])
            sd32 = ref_cpu.make_state_dict(d, meta["weight_seed"], torch.float32, lm_head_gain=meta["gain"], norm_jitter=meta["norm_jitter"])
            models[mk] = (d, {k: v.to(torch.bfloat16) for k, v in sd32.items()})
        d, sd = models[mk]
        tr = ref_cpu.diffusion_generate(sd, d, torch.from_numpy(separated[key + "_ids"]), max_new_tokens=meta["G"],
                                        steps=meta["steps"], alg=meta["alg"], block_length=meta.get("block_length"))
        assert np.array_equal(tr.sequences.numpy(), separated[key + "_seq"]), key
        assert np.array_equal(torch.stack(tr.history).numpy(), separated[key + "_hist"]), key
