"""CPU oracle for CoDA's masked-diffusion sampler.  TEST INFRASTRUCTURE ONLY.

This file is a CPU (torch, no GPU) restatement of the algorithm behind
``CoDALanguageModel.diffusion_generate``.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / CPU baseline.  The shipped package
(``coda_amd``) never imports anything from ``oracle/``.

Parity status: PINNED.  ``tests/golden/make_golden.py`` (run in the build
container, where the read-only reference is importable) drives the reference
sampler and this restatement on the same weights, inputs and seeds and stores
the reference's outputs as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks this file against those fixtures on
any machine.  The reference ships no tests or golden vectors of its own for
this path.

Reference lines restated (paths relative to the reference checkout):
  forward      CoDALanguageModel/modeling_coda.py:36-41,58-60,114-156,176-197,
               216-255,312-355,406-412; attention.py:12-90;
               modeling_utils.py:54-101
  sampler      CoDALanguageModel/generation_utils.py:39-58 (top-p / top-k),
               61-100 (sample_tokens), 412-552 (_sample)
  block_length SURVEY.md appendix D (build-defined windowed _sample; reduces to
               the reference when block_length is None or == max_new_tokens)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- dims / weights
@dataclass
class Dims:
    """Model dimensions (defaults = CoDA-v0 1.7B, model_config.py:15-39)."""

    vocab: int = 151936
    hidden: int = 2048
    inter: int = 6144
    heads: int = 16
    kv_heads: int = 8
    head_dim: int = 128
    layers: int = 28
    rms_eps: float = 1e-6
    rope_theta: float = 1000000
    mask_id: int = 151669
    pad_id: int = 151643
    max_pos: int = 40960

    def shapes(self) -> Dict[str, tuple]:
        """State-dict keys and shapes."""
        s: Dict[str, tuple] = {"model.embed_tokens.weight": (self.vocab, self.hidden)}
        for i in range(self.layers):
            p = f"model.layers.{i}."
            s[p + "self_attn.q_proj.weight"] = (self.heads * self.head_dim, self.hidden)
            s[p + "self_attn.k_proj.weight"] = (self.kv_heads * self.head_dim, self.hidden)
            s[p + "self_attn.v_proj.weight"] = (self.kv_heads * self.head_dim, self.hidden)
            s[p + "self_attn.o_proj.weight"] = (self.hidden, self.heads * self.head_dim)
            s[p + "self_attn.q_norm.weight"] = (self.head_dim,)
            s[p + "self_attn.k_norm.weight"] = (self.head_dim,)
            s[p + "mlp.gate_proj.weight"] = (self.inter, self.hidden)
            s[p + "mlp.up_proj.weight"] = (self.inter, self.hidden)
            s[p + "mlp.down_proj.weight"] = (self.hidden, self.inter)
            s[p + "input_layernorm.weight"] = (self.hidden,)
            s[p + "post_attention_layernorm.weight"] = (self.hidden,)
        s["model.norm.weight"] = (self.hidden,)
        s["lm_head.weight"] = (self.vocab, self.hidden)
        return s


def make_state_dict(dims: Dims, seed: int, dtype=torch.float32, lm_head_gain: float = 1.0,
                    norm_jitter: float = 0.0) -> Dict[str, torch.Tensor]:
    """Synthetic weights with the reference's init statistics (modeling_coda.py:379-388):
    normal(0, 0.02) for Linear/Embedding, ones for the norms.  Drawn from a seeded CPU
    generator in canonical key order, so every machine with this torch regenerates the
    same tensors.  ``lm_head_gain`` scales lm_head ("peaked" fixtures);
    ``norm_jitter`` perturbs norm weights so they are exercised by parity tests."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for key, shape in dims.shapes().items():
        if len(shape) == 1:
            w = torch.ones(shape)
            if norm_jitter:
                w = w + norm_jitter * torch.randn(shape, generator=g)
        else:
            w = torch.randn(shape, generator=g) * 0.02
            if key == "lm_head.weight":
                w = w * lm_head_gain
        out[key] = w.to(dtype)
    return out


# --------------------------------------------------------------------------- forward
def _rms(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_tables(L: int, head_dim: int, theta, dtype) -> tuple:
    """cos/sin [L, head_dim], computed in fp32 and cast to the model dtype."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    ang = torch.arange(L).float()[:, None] * inv[None, :]
    emb = torch.cat((ang, ang), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rot_half(t: torch.Tensor) -> torch.Tensor:
    h = t.shape[-1] // 2
    return torch.cat((-t[..., h:], t[..., :h]), dim=-1)


def hidden_states(sd: Dict[str, torch.Tensor], dims: Dims, ids: torch.Tensor) -> torch.Tensor:
    """Final-normed hidden states [B, L, H] (eval branch; attention mask ignored)."""
    B, L = ids.shape
    h = F.embedding(ids, sd["model.embed_tokens.weight"])
    cos, sin = rope_tables(L, dims.head_dim, dims.rope_theta, h.dtype)
    cos, sin = cos[None, None], sin[None, None]
    rep = dims.heads // dims.kv_heads
    for i in range(dims.layers):
        p = f"model.layers.{i}."
        a = _rms(h, sd[p + "input_layernorm.weight"], dims.rms_eps)
        q = F.linear(a, sd[p + "self_attn.q_proj.weight"]).view(B, L, dims.heads, dims.head_dim)
        k = F.linear(a, sd[p + "self_attn.k_proj.weight"]).view(B, L, dims.kv_heads, dims.head_dim)
        v = F.linear(a, sd[p + "self_attn.v_proj.weight"]).view(B, L, dims.kv_heads, dims.head_dim)
        q = _rms(q, sd[p + "self_attn.q_norm.weight"], dims.rms_eps).transpose(1, 2)
        k = _rms(k, sd[p + "self_attn.k_norm.weight"], dims.rms_eps).transpose(1, 2)
        v = v.transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        if rep > 1:
            k = k.repeat_interleave(rep, dim=1)
            v = v.repeat_interleave(rep, dim=1)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, L, dims.heads * dims.head_dim)
        h = h + F.linear(o, sd[p + "self_attn.o_proj.weight"])
        a = _rms(h, sd[p + "post_attention_layernorm.weight"], dims.rms_eps)
        gate = F.linear(a, sd[p + "mlp.gate_proj.weight"])
        up = F.linear(a, sd[p + "mlp.up_proj.weight"])
        h = h + F.linear(F.silu(gate) * up, sd[p + "mlp.down_proj.weight"])
    return _rms(h, sd["model.norm.weight"], dims.rms_eps)


def forward_logits(sd, dims: Dims, ids: torch.Tensor) -> torch.Tensor:
    """Unshifted logits [B, L, V] (what ``model(x)`` returns in eval mode)."""
    return F.linear(hidden_states(sd, dims, ids), sd["lm_head.weight"])


# --------------------------------------------------------------------------- noise plumbing
class TorchNoise:
    """Draws sampler noise from torch's global CPU generator in the reference's order and
    shapes, so a seeded oracle run equals a seeded reference run on the CPU.
    Every draw is also recorded so a test can replay the same noise into the HIP path."""

    def __init__(self):
        self.log: List[tuple] = []

    def token_exp(self, step: int, probs: torch.Tensor) -> torch.Tensor:
        q = torch.empty_like(probs).exponential_(1)
        self.log.append(("token", step, q))
        return q

    def select_exp(self, step: int, probs: torch.Tensor) -> torch.Tensor:
        q = torch.empty_like(probs).exponential_(1)
        self.log.append(("select", step, q))
        return q

    def origin_uniform(self, step: int, n: int) -> torch.Tensor:
        u = torch.rand(n)
        self.log.append(("origin", step, u))
        return u


class ReplayNoise:
    """Feeds back noise recorded by ``TorchNoise`` (or any list of (kind, step, tensor))."""

    def __init__(self, log):
        self._by = {(k, s): t for k, s, t in log}

    def token_exp(self, step, probs):
        return self._by[("token", step)].to(probs.dtype)

    def select_exp(self, step, probs):
        return self._by[("select", step)].to(probs.dtype)

    def origin_uniform(self, step, n):
        return self._by[("origin", step)]


# --------------------------------------------------------------------------- sample_tokens
def top_p_filter(logits: torch.Tensor, top_p: float) -> torch.Tensor:
    srt, order = torch.sort(logits, descending=True)
    cum = torch.cumsum(F.softmax(srt, dim=-1), dim=-1)
    drop = cum > top_p
    drop[..., 1:] = drop[..., :-1].clone()      # keep the first token that crosses top_p
    drop[..., 0] = False
    mask = torch.zeros_like(logits, dtype=torch.bool).scatter_(-1, order, drop)
    return logits.masked_fill(mask, torch.finfo(logits.dtype).min)


def top_k_filter(logits: torch.Tensor, top_k: int) -> torch.Tensor:
    k = min(top_k, logits.size(-1))
    kth = torch.topk(logits, k)[0][..., -1, None]
    return logits.masked_fill(logits < kth, torch.finfo(logits.dtype).min)


def sample_tokens(logits: torch.Tensor, temperature: float = 0.0, top_p=None, top_k=None,
                  margin: bool = False, neg_entropy: bool = False, noise=None, step: int = 0, diag=None):
    """Returns (confidence[N], x0[N]) for rows of logits [N, V].  ``diag`` (a dict) receives
    'gap': the relative distance between the winning and the runner-up token of every row (used by
    parity tests to tell a genuine near-tie from a wrong answer)."""
    if temperature > 0:
        logits = logits / temperature
    if top_p is not None and top_p < 1:
        if diag is not None and logits.shape[0]:
            # mass that lies BEFORE every token in the nucleus order (same sort as top_p_filter): lets a parity test tell
            # whether a token's membership of the nucleus is decided by less than the stated logit tolerance
            srt, order = torch.sort(logits, descending=True)
            ps = F.softmax(srt.float(), dim=-1)
            diag["nucleus_excl"] = torch.zeros_like(ps).scatter_(-1, order, torch.cumsum(ps, dim=-1) - ps)
        logits = top_p_filter(logits, top_p)
    if top_k is not None:
        logits = top_k_filter(logits, top_k)
    probs = torch.softmax(logits, dim=-1)
    if temperature > 0:
        # Categorical(probs).sample() == argmax(probs / Exp(1))  (exponential race)
        q = noise.token_exp(step, probs)
        x0 = (probs / q).argmax(dim=-1)
        conf = probs.gather(-1, x0[:, None]).squeeze(-1)
        if diag is not None and probs.shape[0]:
            t2 = torch.topk((probs / q).float(), 2, dim=-1)[0]
            diag["gap"] = (t2[:, 0] - t2[:, 1]) / t2[:, 0].clamp_min(1e-30)
    else:
        conf, x0 = probs.max(dim=-1)
        if diag is not None and probs.shape[0]:
            t2 = torch.topk(logits.float(), 2, dim=-1)[0]
            diag["gap"] = (t2[:, 0] - t2[:, 1]) / t2[:, 0].abs().clamp_min(1e-30)
            diag["gap_abs"] = t2[:, 0] - t2[:, 1]          # distance of the two largest logits, in logit units
    if diag is not None and probs.shape[0]:
        live = logits.float()
        live = live.masked_fill(live <= torch.finfo(logits.dtype).min, 0.0)     # entries removed by top-p / top-k
        diag["absmax"] = live.abs().max(dim=-1)[0]
        diag["pmax"] = probs.float().max(dim=-1)[0]
        # sensitivity of the confidence to the softmax inputs: sum_v |d conf / d l_v| (fp32, from this row's own
        # probabilities).  A perturbation of the inputs bounded by u / 2 around a common shift (softmax is shift
        # invariant) moves the confidence by at most (u / 2) * sens to first order - the parity tests' tight tolerance.
        p32 = torch.softmax(logits.float(), dim=-1)
        if neg_entropy:
            lp = torch.log(p32.clamp_min(1e-38))
            h32 = (p32 * lp).sum(-1, keepdim=True)
            diag["sens"] = (p32 * (lp - h32).abs()).sum(-1)
        elif margin:
            t2p = torch.topk(p32, 2, dim=-1)[0]
            diag["sens"] = 2 * t2p[:, 0] * (1 - t2p[:, 0]) + 2 * t2p[:, 1] * (1 - t2p[:, 1])
        else:
            px = p32.gather(-1, x0[:, None]).squeeze(-1)
            diag["sens"] = 2 * px * (1 - px)
    if margin:
        top2 = torch.sort(probs, dim=-1, descending=True)[0][:, :2]
        conf = top2[:, 0] - top2[:, 1]
    if neg_entropy:
        conf = torch.sum(probs * torch.log(probs + 1e-10), dim=-1)
    return conf, x0


# --------------------------------------------------------------------------- schedule
def transfer_counts(mask_total: int, batch: int, steps: int, eps: float) -> List[int]:
    """k_i when every step removes exactly k_i masks per row (known answers)."""
    ts = torch.linspace(1, eps, steps + 1)
    out, m = [], torch.tensor(mask_total)
    for i in range(steps):
        nm = m / batch
        k = int(nm * (1 - ts[i + 1] / ts[i])) if i < steps - 1 else int(nm)
        out.append(k)
        m = m - k * batch
    return out


# --------------------------------------------------------------------------- the sampler loop
@dataclass
class SamplerArgs:
    max_length: int
    steps: int
    mask_id: int
    eps: float = 1e-3
    alg: str = "origin"
    alg_temp: Optional[float] = None
    temperature: float = 0.0
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    block_length: Optional[int] = None
    output_history: bool = False
    threshold: float = 0.9          # serving sampler only (alg == "confidence_threshold")


@dataclass
class SamplerTrace:
    sequences: torch.Tensor = None
    history: List[torch.Tensor] = field(default_factory=list)
    # per step: confidence of masked candidates (row-major order), their x0, k and the chosen (row, col) pairs
    conf: List[torch.Tensor] = field(default_factory=list)
    x0: List[torch.Tensor] = field(default_factory=list)
    k: List[int] = field(default_factory=list)
    chosen: List[torch.Tensor] = field(default_factory=list)
    # per step: selection key [B, L] (confidence, or softmax(conf/alg_temp)/noise; -inf off-candidate),
    # candidate mask [B, L] and the relative top-2 gap of every candidate's token choice
    key: List[torch.Tensor] = field(default_factory=list)
    cand: List[torch.Tensor] = field(default_factory=list)
    gap: List[torch.Tensor] = field(default_factory=list)
    absmax: List[torch.Tensor] = field(default_factory=list)     # largest |logit| of every candidate row
    nucleus: List[Optional[torch.Tensor]] = field(default_factory=list)   # top-p only: [n, V] mass before each token
    pmax: List[torch.Tensor] = field(default_factory=list)       # largest probability of every candidate row
    gap_abs: List[Optional[torch.Tensor]] = field(default_factory=list)   # greedy: top-2 logit distance, logit units
    sens: List[Optional[torch.Tensor]] = field(default_factory=list)      # sum_v |d conf / d logit_v| of every row


def sample_loop(logits_fn: Callable[[torch.Tensor], torch.Tensor], input_ids: torch.Tensor,
                args: SamplerArgs, noise=None, tokens_hook=None, logits_hook=None,
                trace: bool = False, force_states: Optional[List[torch.Tensor]] = None,
                max_steps: Optional[int] = None, step_indices: Optional[List[int]] = None) -> SamplerTrace:
    """``logits_fn(x) -> [B, L, V]`` unshifted logits.  ``force_states[i]`` (optional) replaces
    ``x`` at the start of step i (teacher forcing, used by per-step parity checks).  With
    ``step_indices`` only those steps are walked: ``force_states[n]`` is then the state at the start of
    step ``step_indices[n]`` (a bounded sample of a full-size run)."""
    noise = noise or TorchNoise()
    B, P = input_ids.shape
    L = args.max_length
    G = L - P
    x = F.pad(input_ids, (0, G), value=args.mask_id)
    tr = SamplerTrace()

    if args.block_length is None or args.block_length == G:
        n_blocks, bl, steps_b = 1, G, args.steps
    else:
        bl = args.block_length
        if G % bl or args.steps % (G // bl):
            raise ValueError("block_length must divide max_new_tokens and steps must be a multiple of the block count")
        n_blocks, steps_b = G // bl, args.steps // (G // bl)
    ts = torch.linspace(1, args.eps, steps_b + 1)

    if tokens_hook is not None:
        x = tokens_hook(None, x, None)
    steps = args.steps
    if args.alg == "confidence_threshold":
        # serving/dream/model/generation_utils.py:416-422
        assert n_blocks == 1
        mask_index = x == args.mask_id
        assert mask_index.sum() % steps == 0, "mask_index.sum() must be divisible by steps"
        assert x.shape[0] == 1, "batch size must be 1"
        number_transfer_tokens = mask_index.sum().item() // steps
        left_tokens_last_step = 0
    if step_indices is not None:
        assert force_states is not None and len(step_indices) == len(force_states)
    n = -1
    # teacher forcing walks exactly the given states (the threshold alg's own step count is data dependent)
    while n + 1 < (steps if force_states is None else len(force_states)):
        n += 1
        i = n if step_indices is None else step_indices[n]
        if max_steps is not None and i >= max_steps:      # bounded sample of the loop (bench.py cpu_baseline)
            break
        if force_states is not None:
            x = force_states[n].clone()
        blk, j = divmod(min(i, args.steps - 1), steps_b)
        lo, hi = P + blk * bl, P + (blk + 1) * bl
        mask_index = x == args.mask_id
        if n_blocks > 1:
            win = torch.zeros_like(mask_index)
            win[:, lo:hi] = True
            mask_index = mask_index & win
        logits = logits_fn(x)
        logits = torch.cat([logits[:, :1], logits[:, :-1]], dim=1)
        if logits_hook is not None:
            logits = logits_hook(i, x, logits)
        rows = logits[mask_index]
        t, s = ts[j], ts[j + 1]
        last = j == steps_b - 1
        if args.alg == "origin":
            p_transfer = 1 - s / t if not last else 1
            x0 = torch.full((rows.shape[0],), args.mask_id, dtype=torch.long)
            pick = noise.origin_uniform(i, rows.shape[0]) < p_transfer
            dg = {} if trace else None
            _, x0[pick] = sample_tokens(rows[pick], args.temperature, args.top_p, args.top_k, noise=noise, step=i, diag=dg)
            x[mask_index] = x0.clone()
            if trace:
                gap = torch.full((rows.shape[0],), float("inf"))
                if "gap" in dg:
                    gap[pick] = dg["gap"]
                tr.x0.append(x0.clone()); tr.gap.append(gap); tr.cand.append(mask_index.clone())
                tr.key.append(None); tr.k.append(None); tr.absmax.append(None); tr.conf.append(None); tr.pmax.append(None)
                tr.nucleus.append(None); tr.gap_abs.append(None); tr.sens.append(None)
        elif args.alg == "confidence_threshold":
            # serving/dream/model/generation_utils.py:442-465 (the serving sampler bound onto the CoDA model by
            # serving/fast-api/app/core/service.py:99-120)
            dg = {} if trace else None
            conf, x0 = sample_tokens(rows, args.temperature, args.top_p, args.top_k, noise=noise, step=i, diag=dg)
            cand = torch.full_like(x, args.mask_id)
            cand[mask_index] = x0.clone()
            full = torch.full(x.shape, -torch.inf, dtype=logits.dtype)
            full[mask_index] = conf
            current = number_transfer_tokens + left_tokens_last_step
            if force_states is not None:
                current = min(current, int(mask_index.sum()))      # forced states need not follow this run's own counts
            left_tokens_last_step = 0
            sel_conf, sel_idx = torch.topk(full, current)
            transfer = torch.zeros_like(x, dtype=torch.bool)
            transfer[0, sel_idx[0]] = True
            for kk in range(1, current):
                if sel_conf[0, kk] < args.threshold:
                    if i < steps - 1:
                        left_tokens_last_step += 1
                        transfer[0, sel_idx[0, kk]] = False
                    else:
                        number_transfer_tokens = 0
                        steps += 1
                        left_tokens_last_step += 1
                        transfer[0, sel_idx[0, kk]] = False
            x[transfer] = cand[transfer].clone()
            if trace:
                tr.conf.append(conf.clone()); tr.x0.append(x0.clone()); tr.k.append(current)
                tr.chosen.append(transfer.nonzero()); tr.key.append(full.float()); tr.cand.append(mask_index.clone())
                tr.gap.append(dg.get("gap", torch.zeros(0))); tr.absmax.append(dg.get("absmax", torch.zeros(0)))
                tr.pmax.append(dg.get("pmax", torch.zeros(0))); tr.nucleus.append(dg.get("nucleus_excl"))
                tr.gap_abs.append(dg.get("gap_abs")); tr.sens.append(dg.get("sens"))
        else:
            if args.alg not in ("maskgit_plus", "topk_margin", "entropy"):
                raise RuntimeError(f"Unknown alg: {args.alg}")
            dg = {} if trace else None
            conf, x0 = sample_tokens(rows, args.temperature, args.top_p, args.top_k,
                                     margin=args.alg == "topk_margin", neg_entropy=args.alg == "entropy",
                                     noise=noise, step=i, diag=dg)
            n_mask = mask_index.sum() / mask_index.shape[0]
            k = int(n_mask * (1 - s / t)) if not last else int(n_mask)
            full = torch.full(x.shape, -torch.inf, dtype=logits.dtype)
            full[mask_index] = conf
            chosen = torch.zeros((0, 2), dtype=torch.long)
            sel_key = full
            if k > 0:
                if args.alg_temp is None or args.alg_temp == 0:
                    _, idx = torch.topk(full, k)
                else:
                    pr = F.softmax(full / args.alg_temp, dim=-1)
                    q = noise.select_exp(i, pr)
                    sel_key = pr / q
                    _, idx = torch.topk(sel_key, k)     # == torch.multinomial(pr, k) without replacement
                cand = torch.full_like(x, args.mask_id)
                cand[mask_index] = x0.clone()
                r = torch.arange(B)[:, None].expand_as(idx)
                x[r, idx] = cand[r, idx]
                chosen = torch.stack([r.reshape(-1), idx.reshape(-1)], dim=1)
            if trace:
                tr.conf.append(conf.clone()); tr.x0.append(x0.clone()); tr.k.append(k); tr.chosen.append(chosen)
                tr.key.append(sel_key.float().masked_fill(~mask_index, -torch.inf)); tr.cand.append(mask_index.clone())
                tr.gap.append(dg.get("gap", torch.zeros(0))); tr.absmax.append(dg.get("absmax", torch.zeros(0)))
                tr.pmax.append(dg.get("pmax", torch.zeros(0))); tr.nucleus.append(dg.get("nucleus_excl"))
                tr.gap_abs.append(dg.get("gap_abs")); tr.sens.append(dg.get("sens"))
        if tokens_hook is not None:
            x = tokens_hook(i, x, logits)
        if args.output_history or trace:
            tr.history.append(x.clone())
    tr.sequences = x
    return tr


def diffusion_generate(sd, dims: Dims, input_ids: torch.Tensor, *, max_new_tokens: int, steps: int,
                       alg: str = "origin", alg_temp=None, temperature: float = 0.0, top_p=None, top_k=None,
                       eps: float = 1e-3, block_length=None, noise=None, trace: bool = False,
                       tokens_hook=None, logits_hook=None, force_states=None, max_steps=None,
                       threshold: float = 0.9, step_indices=None) -> SamplerTrace:
    """Oracle twin of ``model.diffusion_generate`` for equal-length prompts."""
    args = SamplerArgs(max_length=input_ids.shape[1] + max_new_tokens, steps=steps, mask_id=dims.mask_id, eps=eps,
                       alg=alg, alg_temp=alg_temp, temperature=temperature, top_p=top_p, top_k=top_k,
                       block_length=block_length, output_history=True, threshold=threshold)
    with torch.no_grad():
        return sample_loop(lambda x: forward_logits(sd, dims, x), input_ids, args, noise=noise,
                           tokens_hook=tokens_hook, logits_hook=logits_hook, trace=trace, force_states=force_states,
                           max_steps=max_steps, step_indices=step_indices)


# --------------------------------------------------------------------------- lm-eval adapter (id level)
def generate_batch_ids(generate_fn, prompts, *, pad_id: int, eos_id: int, max_prompt_len: int) -> List[List[int]]:
    """Id-level restatement of ``DiffLLM._generate_batch`` (evaluation/lm_eval/lm_eval/models/diffllm.py:265-295).

    ``generate_fn(prompt_ids[B, P], attn_mask[B, P]) -> sequences[B, L]`` stands for the
    ``model.diffusion_generate(...)`` call of :275-287.  The tokenizer steps become: left padding with
    ``pad_id`` (:267), the ``[:, -max_prompt_len:]`` slice (:268), the mask ``ids != pad`` (:269), and
    ``decode(g[len(p):]).split(eos_token)[0]`` (:291) = the ids before the first ``eos_id``.
    Parity of these six lines is by restatement only (the module cannot be imported here: its package
    needs ``sacrebleu``); the ``diffusion_generate`` call on left-padded prompts is pinned by the
    ``lmeval_*`` golden vectors."""
    width = max(len(p) for p in prompts)
    ids = torch.full((len(prompts), width), pad_id, dtype=torch.long)
    for r, p in enumerate(prompts):
        if len(p):
            ids[r, width - len(p):] = torch.tensor(list(p), dtype=torch.long)
    ids = ids[:, -max_prompt_len:]
    attn = ids.ne(pad_id)
    seqs = generate_fn(ids, attn)
    out = []
    for p, g in zip(ids, seqs):
        tail = g[len(p):].tolist()
        out.append(tail[:tail.index(eos_id)] if eos_id in tail else tail)
    return out


# --------------------------------------------------------------------------- online forms
def online_confidence(logits: torch.Tensor, alg: str, temperature: float = 0.0, q: Optional[torch.Tensor] = None,
                      tile: int = 128):
    """fp64 single-pass (tile-by-tile) evaluation of the confidence reductions carried by the
    fused lm-head epilogue; used to bound the kernel's algorithmic (not rounding) error."""
    l = logits.double()
    if temperature > 0:
        l = l / temperature
    N, V = l.shape
    m = torch.full((N,), -math.inf, dtype=torch.float64)
    m2 = torch.full((N,), -math.inf, dtype=torch.float64)
    Z = torch.zeros(N, dtype=torch.float64)
    S = torch.zeros(N, dtype=torch.float64)
    arg = torch.zeros(N, dtype=torch.long)
    arg_s = torch.zeros(N, dtype=torch.long)
    best_key = torch.full((N,), -math.inf, dtype=torch.float64)
    best_l = torch.zeros(N, dtype=torch.float64)
    for v0 in range(0, V, tile):
        t = l[:, v0:v0 + tile]
        tm, ta = t.max(dim=1)
        top2 = torch.topk(torch.cat([t, m[:, None], m2[:, None]], dim=1), 2, dim=1)[0]
        newm = torch.maximum(m, tm)
        scale = torch.exp(m - newm)
        scale[torch.isinf(m)] = 0.0
        S = scale * (S + (m - newm).nan_to_num(0.0, 0.0, 0.0) * Z)
        Z = Z * scale
        e = torch.exp(t - newm[:, None])
        Z = Z + e.sum(1)
        S = S + (e * (t - newm[:, None])).sum(1)
        upd = tm > m
        arg = torch.where(upd, ta + v0, arg)
        m, m2 = top2[:, 0], top2[:, 1]
        if q is not None:
            key = t - torch.log(q[:, v0:v0 + tile].double())
            km, ka = key.max(dim=1)
            better = km > best_key
            best_key = torch.where(better, km, best_key)
            best_l = torch.where(better, t.gather(1, ka[:, None]).squeeze(1), best_l)
            arg_s = torch.where(better, ka + v0, arg_s)
    if temperature > 0 and q is not None:
        pmax = torch.exp(best_l - m) / Z
        arg = arg_s
    else:
        pmax = 1.0 / Z
    if alg == "maskgit_plus":
        conf = pmax
    elif alg == "topk_margin":
        conf = (1.0 - torch.exp(m2 - m)) / Z
    elif alg == "entropy":
        conf = S / Z - torch.log(Z)
    else:
        raise RuntimeError(f"Unknown alg: {alg}")
    return conf, arg
