"""Id-level mirror of the reference's lm-eval adapter for the sampler path.

Reference: ``evaluation/lm_eval/lm_eval/models/diffllm.py`` - the ``DiffLLM`` constructor's generation
parameters (:46-66, :174-183), ``_generate_batch`` (:265-295) and the batching loop of ``generate_until``
(:297-322).  The reference works on strings through a tokenizer that is not part of the checkout
(``tokenizer.json`` is a missing large blob), so this adapter starts and ends at token ids:

  text step in the reference                               id-level equivalent here
  -------------------------------------------------------  ------------------------------------------------
  tokenizer(prompts, padding=True, padding_side="left")    left-pad the id lists with ``pad_token_id``
  prompt_ids[:, -max_prompt_len:]                          same slice
  attn_mask = prompt_ids.ne(pad)                           same (the model accepts and ignores it)
  decode(g[len(p):]).split(eos_token)[0]                   ids of g[len(p):] before the first ``eos_token_id``
  r.split(s)[0] for s in until                             cut before the first occurrence of an ``until`` id sequence

Everything between is the drop-in ``model.diffusion_generate`` with the adapter's keyword set
(``return_dict_in_generate=True, output_history=False``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def left_pad(prompts: Sequence[Sequence[int]], pad_token_id: int) -> torch.Tensor:
    """``tokenizer(..., padding=True, padding_side="left").input_ids`` on already-tokenised prompts."""
    if len(prompts) == 0:
        raise ValueError("empty batch")
    width = max(len(p) for p in prompts)
    if width == 0:
        raise ValueError("all prompts are empty")
    ids = torch.full((len(prompts), width), pad_token_id, dtype=torch.long)
    for r, p in enumerate(prompts):
        if len(p):
            ids[r, width - len(p):] = torch.as_tensor(list(p), dtype=torch.long)
    return ids


def cut_at(tokens: List[int], stop: Sequence[int]) -> List[int]:
    """``text.split(stop)[0]`` on id sequences: everything before the first occurrence of ``stop``."""
    n = len(stop)
    if n == 0:
        return tokens
    for i in range(len(tokens) - n + 1):
        if tokens[i:i + n] == list(stop):
            return tokens[:i]
    return tokens


class DiffLLMAdapter:
    """Generation half of the reference's ``DiffLLM`` (diffllm.py:46-322) on token ids.

    ``model`` is a ``coda_amd.CoDALanguageModel`` on a GPU.  Defaults are the reference's: ``batch_size=1``,
    ``max_prompt_len=1024``, ``max_new_tokens=128``, ``diffusion_steps=32``, ``temperature=0.1``, ``top_p=0.9``,
    ``alg="entropy"``, ``alg_temp=0.0``, ``top_k=None``.
    """

    def __init__(self, model, pad_token_id: Optional[int] = None, eos_token_id: Optional[int] = None, batch_size: int = 1,
                 max_prompt_len: int = 1024, max_new_tokens: int = 128, diffusion_steps: int = 32, rank: int = 0,
                 world_size: int = 1, **kwargs):
        self.model = model
        cfg = getattr(model, "config", None)
        self.pad_token_id = pad_token_id if pad_token_id is not None else getattr(cfg, "pad_token_id", None)
        self.eos_token_id = eos_token_id if eos_token_id is not None else getattr(cfg, "eos_token_id", None)
        if self.pad_token_id is None or self.eos_token_id is None:
            raise ValueError("pad_token_id and eos_token_id are required (the tokenizer supplies them in the reference)")
        self.batch_size_per_gpu = int(batch_size)
        self.max_prompt_len = max_prompt_len
        self.max_new_tokens = max_new_tokens
        self.diffusion_steps = diffusion_steps
        self.temperature = kwargs.get("temperature", 0.1)
        self.top_p = kwargs.get("top_p", 0.9)
        self.alg = kwargs.get("alg", "entropy")
        self.alg_temp = kwargs.get("alg_temp", 0.0)
        self.top_k = kwargs.get("top_k", None)
        self.block_length = kwargs.get("block_length", None)      # build-defined extension, None = reference loop
        self._rank, self._world_size = rank, world_size

    @property
    def batch_size(self):
        return self.batch_size_per_gpu

    @property
    def device(self):
        return self.model.device

    @property
    def rank(self):
        return self._rank

    @property
    def world_size(self):
        return self._world_size

    def prepare_batch(self, prompts: Sequence[Sequence[int]]):
        """diffllm.py:267-271 - padded ids and the (ignored) attention mask, on the model's device."""
        prompt_ids = left_pad(prompts, self.pad_token_id)[:, -self.max_prompt_len:]
        attn_mask = prompt_ids.ne(self.pad_token_id)
        return prompt_ids.to(self.device), attn_mask.to(self.device)

    def generation_kwargs(self):
        """diffllm.py:275-287"""
        kw = dict(max_new_tokens=self.max_new_tokens, output_history=False, return_dict_in_generate=True,
                  steps=self.diffusion_steps, temperature=self.temperature, top_p=self.top_p, top_k=self.top_k,
                  alg=self.alg, alg_temp=self.alg_temp)
        if self.block_length is not None:
            kw["block_length"] = self.block_length
        return kw

    def generate_batch_ids(self, prompts: Sequence[Sequence[int]]) -> List[List[int]]:
        prompt_ids, attn_mask = self.prepare_batch(prompts)
        out = self.model.diffusion_generate(prompt_ids, attention_mask=attn_mask, **self.generation_kwargs())
        seqs = out.sequences.cpu()
        # diffllm.py:290-293: the continuation starts after the PADDED prompt width, ends before the first eos
        return [cut_at(g[len(p):].tolist(), [self.eos_token_id]) for p, g in zip(prompt_ids, seqs)]

    def generate_until_ids(self, requests: Sequence[Sequence[int]], until: Sequence[Sequence[int]] = ()) -> List[List[int]]:
        """diffllm.py:297-322 - fixed-size batches in request order; ``until`` (id sequences) is applied to every
        response, in order, like the reference applies the first request's ``until`` strings."""
        res: List[List[int]] = []
        for b0 in range(0, len(requests), self.batch_size):
            responses = self.generate_batch_ids(requests[b0:b0 + self.batch_size])
            for i, r in enumerate(responses):
                for s in until:
                    r = cut_at(r, s)
                responses[i] = r
            res.extend(responses)
        return res
