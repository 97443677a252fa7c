"""Masked-diffusion generation API: ``model.diffusion_generate(...)``.

Drop-in for the reference's DLMGenerationMixin (CoDALanguageModel/generation_utils.py:103-552):
same call signature, same DLMGenerationConfig fields and defaults, same errors and warnings,
same outputs - but the denoise loop runs on hand-written CDNA4 kernels (coda_amd/csrc) and stays
device-resident: no host sync per step, no [B,L,V] logits unless a hook asks for them.

Build-defined additions (all optional, defaults reproduce the reference):
  block_length   windowed unmasking; None or == max_new_tokens -> reference
  round_conf_to  "bf16" rounds confidences to bf16 as the reference's bf16 arithmetic does (F5)
  noise_source   object supplying sampling noise (tests inject recorded noise; default: in-kernel counter hash)
"""
from __future__ import annotations

import copy
import warnings
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from transformers import __version__ as _tf_version
from transformers.generation.configuration_utils import GenerationConfig
from transformers.utils import ModelOutput, logging

from . import _lib

logger = logging.get_logger(__name__)

_FIELDS = dict(temperature=0.0, top_p=None, top_k=None, max_length=20, max_new_tokens=None, eps=1e-3, steps=512,
               alg="origin", alg_temp=None, num_return_sequences=1, return_dict_in_generate=False, output_history=False,
               mask_token_id=None, pad_token_id=None, bos_token_id=None, eos_token_id=None,
               block_length=None, round_conf_to=None)
_TOKEN_FIELDS = ("bos_token_id", "eos_token_id", "pad_token_id", "mask_token_id")


@dataclass
class DLMModelOutput(ModelOutput):
    sequences: torch.LongTensor = None
    history: Optional[Tuple[torch.FloatTensor]] = None


class DLMGenerationConfig(GenerationConfig):
    """Attribute bag with the reference's fields/defaults (generation_utils.py:109-159); validate() is a no-op there."""

    def __init__(self, **kwargs):
        for name, default in _FIELDS.items():
            setattr(self, name, kwargs.pop(name, default))
        self.generation_kwargs = kwargs.pop("generation_kwargs", {})
        self._from_model_config = kwargs.pop("_from_model_config", False)
        self._commit_hash = kwargs.pop("_commit_hash", None)
        self.transformers_version = kwargs.pop("transformers_version", _tf_version)
        if not self._from_model_config:
            for key, value in kwargs.items():
                setattr(self, key, value)

    def validate(self, *args, **kwargs):
        return None

    def update(self, **kwargs):
        """Overlays known attributes, returns the kwargs that were not consumed."""
        unused = {}
        for key, value in kwargs.items():
            if hasattr(self, key):
                setattr(self, key, value)
            else:
                unused[key] = value
        return unused

    @classmethod
    def from_model_config(cls, model_config):
        picked = {k: getattr(model_config, k) for k in _TOKEN_FIELDS if getattr(model_config, k, None) is not None}
        return cls(_from_model_config=True, **picked)


# ---------------------------------------------------------------------------------- noise
class DeviceNoise:
    """Default sampling noise.  Nothing is drawn by torch inside the loop: token noise (temperature > 0), the selection
    race (alg_temp > 0) and the ``origin`` uniforms are all generated INSIDE the kernels from a counter hash keyed by
    (step seed, flat position, index) - coda_hash_uniform in csrc/coda_common.h.  The run seed is one draw from torch's
    CPU generator per generate, so ``torch.manual_seed`` makes a run reproducible.  (Same-seed equality with the
    reference's own RNG stream is impossible across devices and is not claimed: SURVEY.md F6.)"""

    def __init__(self, device=None):
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())

    def token_noise(self, step, cap_rows, V, picked=None):
        return None

    def step_seed(self, step):
        return (self.seed + 0x9E3779B97F4A7C15 * (step + 1)) & 0xFFFFFFFFFFFFFFFF

    def select_noise(self, step, B, L):
        return None

    def origin_uniform(self, step, cap_rows):
        return None


class LazyLogits:
    """What a TOKENS hook receives as its ``logits`` argument when no logits hook forces them to exist.  The reference
    hands every tokens hook the step's shifted [B, L, V] logits (generation_utils.py:541); most hooks only look at the
    step index and ``x``.  This object stands for that tensor and materialises it - one full forward of the state the
    step STARTED from + the shifted lm-head, into the engine's own buffer - the first time the hook touches it (an
    attribute, an index, an arithmetic operator, or passing it to a torch function); a hook that never does leaves the
    fused, logits-free route untouched.  The materialised tensor lives in an engine buffer that the next step reuses:
    clone it to keep it."""

    def __init__(self, engine, x_before):
        object.__setattr__(self, "_engine", engine)
        object.__setattr__(self, "_x", x_before)
        object.__setattr__(self, "_t", None)

    def materialize(self) -> torch.Tensor:
        if self._t is None:
            B, L = self._x.shape
            self._engine.forward_hidden(self._x)
            object.__setattr__(self, "_t", self._engine.shifted_logits(B, L))
        return self._t

    @property
    def materialized(self) -> bool:
        return self._t is not None

    def __getattr__(self, name):
        return getattr(self.materialize(), name)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __len__(self):
        return self._x.shape[0]

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return f"LazyLogits(materialized={self.materialized}, batch={tuple(self._x.shape)})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def unwrap(a):
            if isinstance(a, LazyLogits):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(unwrap(v) for v in a)
            return a
        return func(*unwrap(args), **{k: unwrap(v) for k, v in (kwargs or {}).items()})


def _lazy_binary(name):
    def op(self, other):
        other = other.materialize() if isinstance(other, LazyLogits) else other
        return getattr(self.materialize(), name)(other)
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "floordiv", "pow", "matmul", "radd", "rsub", "rmul", "rtruediv",
           "lt", "le", "gt", "ge", "eq", "ne"):
    setattr(LazyLogits, f"__{_n}__", _lazy_binary(f"__{_n}__"))
LazyLogits.__neg__ = lambda self: -self.materialize()
LazyLogits.__hash__ = object.__hash__


@dataclass
class _Call:
    """Per-call options that are not generation_config fields (kept off the model object: two generate calls must
    not see each other's state)."""
    threshold: float = 0.9              # serving sampler's keyword (serving/dream/model/generation_utils.py:357)
    debug_trace: Optional[list] = None  # tests: receives per-step device confidences
    prompt_has_mask: bool = False
    noise_source: object = None


def fp32_schedule(steps_per_block: int, eps: float):
    """(1 - s/t) per step in fp32, with the reference's tensor arithmetic (generation_utils.py:450,463-464,509)."""
    ts = torch.linspace(1, eps, steps_per_block + 1)
    return [float(1 - ts[j + 1] / ts[j]) for j in range(steps_per_block)]


def plan_windows(prompt_len: int, gen_len: int, steps: int, block_length):
    """-> (n_blocks, block_len, steps_per_block); reference mode is one block spanning the row."""
    if block_length is None or block_length == gen_len:
        return 1, gen_len, steps
    if block_length <= 0 or gen_len % block_length:
        raise ValueError(f"block_length={block_length} must divide max_new_tokens={gen_len}")
    n_blocks = gen_len // block_length
    if steps % n_blocks:
        raise ValueError(f"steps={steps} must be a multiple of the number of blocks ({n_blocks})")
    return n_blocks, block_length, steps // n_blocks


class DLMGenerationMixin:
    @staticmethod
    def _expand_inputs_for_generation(expand_size=1, input_ids=None, attention_mask=None):
        if expand_size == 1:
            return input_ids, attention_mask
        if input_ids is not None:
            input_ids = input_ids.repeat_interleave(expand_size, dim=0)
        if attention_mask is not None:
            attention_mask = attention_mask.repeat_interleave(expand_size, dim=0)
        return input_ids, attention_mask

    def _prepare_generation_config(self, generation_config, **kwargs):
        from_model = generation_config is None
        if from_model:
            generation_config = DLMGenerationConfig.from_model_config(self.config)
        generation_config = copy.deepcopy(generation_config)
        generation_config.update(**kwargs)
        for name in _FIELDS:                         # configs built by other code may lack the added fields
            if not hasattr(generation_config, name):
                setattr(generation_config, name, _FIELDS[name])
        if not from_model:
            for name in _TOKEN_FIELDS:
                if getattr(generation_config, name, None) is None:
                    setattr(generation_config, name, getattr(self.generation_config, name, None))
        if generation_config.mask_token_id is None:
            generation_config.mask_token_id = getattr(self.config, "mask_token_id", None)
        return generation_config

    def _prepare_generated_length(self, generation_config, has_default_max_length, input_ids_length):
        if generation_config.max_new_tokens is not None:
            if not has_default_max_length and generation_config.max_length is not None:
                logger.warning(
                    f"Both `max_new_tokens` (={generation_config.max_new_tokens}) and `max_length`(="
                    f"{generation_config.max_length}) seem to have been set. `max_new_tokens` will take precedence.")
            generation_config.max_length = generation_config.max_new_tokens + input_ids_length
        elif has_default_max_length and generation_config.max_length == _FIELDS["max_length"]:
            generation_config.max_length = generation_config.max_length + input_ids_length
            cap = getattr(self.config, "max_position_embeddings", None)
            if cap is not None:
                generation_config.max_length = min(generation_config.max_length, cap)
        return generation_config

    def _validate_generated_length(self, generation_config, input_ids_length, has_default_max_length):
        if has_default_max_length and generation_config.max_new_tokens is None and generation_config.max_length == 20:
            warnings.warn(
                f"Using the model-agnostic default `max_length` (={generation_config.max_length}) to control the "
                "generation length. We recommend setting `max_new_tokens` to control the maximum length of the generation.",
                UserWarning)
        if input_ids_length >= generation_config.max_length:
            raise ValueError(
                f"Input length of input_ids is {input_ids_length}, but `max_length` is set to"
                f" {generation_config.max_length}. This can lead to unexpected behavior. You should consider"
                " increasing `max_length` or, better yet, setting `max_new_tokens`.")

    @torch.no_grad()
    def diffusion_generate(self, inputs: Optional[torch.Tensor] = None,
                           generation_config: Optional[DLMGenerationConfig] = None, **kwargs):
        tokens_hook = kwargs.pop("generation_tokens_hook_func", None)
        logits_hook = kwargs.pop("generation_logits_hook_func", None)
        attention_mask = kwargs.pop("attention_mask", None)
        call = _Call(noise_source=kwargs.pop("noise_source", None), threshold=kwargs.pop("threshold", 0.9),
                     debug_trace=kwargs.pop("debug_trace", None))
        had_max_length_kwarg = kwargs.get("max_length") is not None
        generation_config = self._prepare_generation_config(generation_config, **kwargs)

        assert inputs is not None
        input_ids = inputs
        self._prepare_special_tokens(generation_config)
        P = input_ids.shape[-1]
        has_default_max_length = (not had_max_length_kwarg) and generation_config.max_length is not None
        generation_config = self._prepare_generated_length(generation_config, has_default_max_length, P)
        self._validate_generated_length(generation_config, P, has_default_max_length)

        if self.device.type != input_ids.device.type:
            warnings.warn(
                "You are calling .generate() with the `input_ids` being on a device type different than your model's"
                f" device. `input_ids` is on {input_ids.device.type}, whereas the model is on {self.device.type}.",
                UserWarning)
        if input_ids.device != self.device:
            # the reference fails inside embed_tokens with a Python device-mismatch error; here the ids' address would go
            # straight into a kernel, so the same condition must be an exception BEFORE any launch, never a GPU fault
            raise _lib.CodaHipError(
                f"input_ids is on {input_ids.device}, but the model is on {self.device}: call input_ids.to('{self.device}') "
                "first (coda_amd has no CPU path and does not move inputs silently)")
        if input_ids.dtype.is_floating_point or input_ids.dtype == torch.bool:
            raise _lib.CodaHipError(f"input_ids must be an integer tensor, got {input_ids.dtype}")
        # One host read before the loop (the reference has it too: torch.any(input_ids == pad), generation_utils.py:386-390).
        # The mask-in-prompt probe and the id range check ride on it: nn.Embedding raises on ids outside [0, V) in the
        # reference, while a gather kernel cannot (it would read another row, or fault).
        pad = generation_config.pad_token_id
        if generation_config.mask_token_id is None:
            raise ValueError("mask_token_id is not set (neither in generation_config nor in the model config)")
        probe = torch.stack([(input_ids == pad).any() if pad is not None else input_ids.new_zeros((), dtype=torch.bool),
                             (input_ids == generation_config.mask_token_id).any(),
                             (input_ids < 0).any() | (input_ids >= self.config.vocab_size).any()]).tolist()
        if probe[2]:
            raise IndexError(f"index out of range in self: input_ids must lie in [0, {self.config.vocab_size})")
        if not 0 <= int(generation_config.mask_token_id) < self.config.vocab_size:
            raise IndexError(f"mask_token_id {generation_config.mask_token_id} is outside the vocabulary [0, {self.config.vocab_size})")
        if probe[0] and attention_mask is None:
            warnings.warn(
                "Padding was detected but no attention mask is passed here. For correct generation results, please "
                "set `attention_mask` when batch-padding inputs.", UserWarning)
        call.prompt_has_mask = bool(probe[1])

        input_ids, attention_mask = self._expand_inputs_for_generation(
            generation_config.num_return_sequences, input_ids, attention_mask)
        return self._sample(input_ids, attention_mask, generation_config, tokens_hook, logits_hook, call)

    def _prepare_special_tokens(self, generation_config):
        """generation_utils.py:279-334.  The reference turns the four ids into tensors kept in private attributes that
        nothing on this path reads; what is observable is the pad <- eos defaulting (:316-325) and its warning."""
        eos = generation_config.eos_token_id
        if isinstance(eos, torch.Tensor):
            eos = eos.reshape(-1).tolist()
        if generation_config.pad_token_id is None and eos is not None:
            first = eos[0] if isinstance(eos, (list, tuple)) else eos
            generation_config.pad_token_id = int(first)
            logger.warning(f"Setting `pad_token_id` to `eos_token_id`:{generation_config.pad_token_id} for open-end generation.")

    @staticmethod
    def _hooked_tokens(x, eng, shape):
        """What a tokens hook returned goes straight into kernels: same checks as the ids at the door."""
        if not isinstance(x, torch.Tensor) or tuple(x.shape) != tuple(shape):
            raise ValueError(f"generation_tokens_hook_func must return a tensor of shape {tuple(shape)}")
        if x.device != eng.device:
            raise _lib.CodaHipError(f"generation_tokens_hook_func returned a tensor on {x.device}, the model is on {eng.device}")
        if x.dtype.is_floating_point or x.dtype == torch.bool:
            raise _lib.CodaHipError(f"generation_tokens_hook_func must return an integer tensor, got {x.dtype}")
        return x.to(torch.int64).contiguous()

    # ------------------------------------------------------------------------------ the denoise loop
    def _sample(self, input_ids, attention_mask, generation_config, tokens_hook, logits_hook, call=None):
        """Device-resident restatement of the reference loop (generation_utils.py:412-552).
        ``attention_mask`` is accepted and has no effect, as in the reference."""
        gc = generation_config
        call = call if call is not None else _Call()
        noise_source = call.noise_source
        alg, alg_temp, temperature = gc.alg, gc.alg_temp, float(gc.temperature or 0.0)
        if alg not in ("origin", "maskgit_plus", "topk_margin", "entropy", "confidence_threshold"):
            raise RuntimeError(f"Unknown alg: {alg}")
        eng = self.hip_engine()
        if input_ids.device != eng.device:
            raise _lib.CodaHipError(f"input_ids is on {input_ids.device}, but the model's HIP engine is on {eng.device}")
        eng.set_mask_id(gc.mask_token_id)
        want_history = bool(gc.return_dict_in_generate and gc.output_history)
        histories = [] if want_history else None

        P = input_ids.shape[1]
        x = F.pad(input_ids, (0, gc.max_length - P), value=gc.mask_token_id).to(torch.int64).contiguous()
        B, L = x.shape
        G = L - P
        n_blocks, bl, steps_b = plan_windows(P, G, gc.steps, gc.block_length)
        ratios = fp32_schedule(steps_b, gc.eps)
        noise = noise_source if noise_source is not None else DeviceNoise(x.device)
        round_conf = gc.round_conf_to in ("bf16", torch.bfloat16, "bfloat16")
        top_p = gc.top_p if (gc.top_p is not None and gc.top_p < 1) else None
        top_k = gc.top_k
        # hooks (generation_utils.py:453,460,541).  A LOGITS hook sees and may rewrite the shifted [B, L, V] logits: they
        # are materialised every step (reference semantics), into one engine-owned buffer, and the candidate rows are
        # compacted from what the hook returned.  A TOKENS hook alone keeps the fused route: it is handed a lazily
        # materialised logits object that costs nothing unless the hook actually reads it.
        materialise_all = logits_hook is not None
        filtered = top_p is not None or top_k is not None
        alg_id = _lib.ALG_IDS.get(alg, 0)                # confidence_threshold scores like maskgit_plus: p[x0]
        a_temp = float(alg_temp) if alg_temp else 0.0

        if tokens_hook is not None:
            x = self._hooked_tokens(tokens_hook(None, x, None), eng, (B, L))
        whole_row = call.prompt_has_mask or tokens_hook is not None     # a tokens hook may re-mask any position
        cap_max = B * ((L if whole_row else G) if n_blocks == 1 else bl)
        eng.reserve(B, L, cap_max, filtered, full_logits=materialise_all)
        steps = gc.steps
        if alg == "confidence_threshold":
            # serving/dream/model/generation_utils.py:416-422: one sequence, a fixed quota per step plus whatever
            # earlier steps held back; the loop grows by one step whenever the last step still holds tokens back
            if n_blocks != 1:
                raise ValueError("confidence_threshold is defined on the reference (non-windowed) loop only")
            n_masks = int((x == gc.mask_token_id).sum().item())
            assert n_masks % steps == 0, "mask_index.sum() must be divisible by steps"
            assert B == 1, "batch size must be 1"
            quota, held_back = n_masks // steps, 0
            threshold = float(call.threshold)
            if round_conf:                                 # bf16 confidences compare against the bf16 threshold
                threshold = float(torch.tensor(threshold, dtype=torch.bfloat16))
        i = -1
        while i + 1 < steps:
            i += 1
            blk, j = divmod(min(i, gc.steps - 1), steps_b)
            if n_blocks == 1:
                # reference mode: every masked position of the row is a candidate (prompt included if it holds masks)
                lo, hi = (0 if whole_row else P), L
            else:
                lo, hi = P + blk * bl, P + (blk + 1) * bl
            cap = B * (hi - lo)
            last = j == steps_b - 1

            u = picked = None
            if alg == "origin":
                p_transfer = 1.0 if last else ratios[j]
                u = noise.origin_uniform(i, cap)               # None: generated in the transfer kernel from the step seed
                picked = (u < p_transfer) if u is not None else None
            tok_noise = noise.token_noise(i, cap, eng.V, picked) if temperature > 0 else None
            seed = noise.step_seed(i)
            logits = None
            if materialise_all:
                eng.forward_hidden(x)                                          # every row: the hook sees [B, L, V]
                logits = logits_hook(i, x, eng.shifted_logits(B, L))          # logits[:, p] predicts token p (:457)
                eng.build_candidates(x, lo, hi)
                eng.confidence_from_full_logits(logits, cap, alg_id, temperature, top_p, top_k, tok_noise, seed, round_conf)
            else:
                # the lm-head reads the hidden state at p-1 for every candidate p in [lo, hi) (:457,462): positions [lo-1, hi-1)
                s_lo = max(lo - 1, 0)
                eng.forward_hidden(x, need=(s_lo, max(hi - 1, s_lo + 1)))
                eng.build_candidates(x, lo, hi)
                eng.lmhead_confidence(cap, alg_id, temperature, tok_noise, seed, round_conf, top_p, top_k)
                if tokens_hook is not None:
                    logits = LazyLogits(eng, x.clone())

            if call.debug_trace is not None:
                call.debug_trace.append(dict(step=i, conf=eng.conf.clone(), x0=eng.x0.clone(), count=eng.count.clone(),
                                              dst=eng.dst_pos.clone(), lo=lo, hi=hi))
            if alg == "origin":
                eng.origin_transfer(x, cap, u, p_transfer, seed)
            elif alg == "confidence_threshold":
                current = quota + held_back
                _, held = eng.select_threshold(x, lo, hi, current, threshold)
                held_back = held
                if held and i >= steps - 1:                # :459-463 - the last step may not leave tokens behind
                    quota = 0
                    steps += 1
            else:
                sel_noise = noise.select_noise(i, B, L) if a_temp > 0 else None
                eng.select_unmask(x, lo, hi, ratios[j], last, a_temp, sel_noise, seed)
                if call.debug_trace is not None:
                    call.debug_trace[-1]["k"] = eng.k_out.clone()

            if tokens_hook is not None:
                x = self._hooked_tokens(tokens_hook(i, x, logits), eng, (B, L))
            if histories is not None:
                histories.append(x.clone())

        if gc.return_dict_in_generate:
            return DLMModelOutput(sequences=x, history=histories)
        return x
