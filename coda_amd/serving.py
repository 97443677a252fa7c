"""Id-level mirror of the reference's serving generation call.

Reference: ``serving/fast-api/app/core/service.py`` binds the serving sampler
(``serving/dream/model/generation_utils.py``: the same denoise loop plus ``alg="confidence_threshold"``) onto the
CoDA model (:75-120) and calls it for one chat request at a time (:205-240) with the defaults of
``serving/fast-api/app/core/config.py:35-42``.  The tokenizer / chat template / HTTP layers are out of scope; this
module starts at prompt token ids and ends at completion token ids.  ``coda_amd.CoDALanguageModel`` already provides
``diffusion_generate(..., threshold=...)`` with the serving sampler's extra algorithm, so nothing is re-bound here.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from .generation_utils import DLMGenerationConfig


@dataclass
class ServingDefaults:
    """serving/fast-api/app/core/config.py:35-42"""
    max_tokens: int = 256
    temperature: float = 0.0
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    steps: int = 128
    alg: str = "entropy"
    alg_temp: float = 0.1
    block_length: int = 32          # read by the service but never passed to the sampler (service.py:197,211-224)


class SamplerService:
    """``Service.generate_chat_completion`` (service.py:168-262) from the formatted prompt ids onward."""

    def __init__(self, model, mask_token_id=None, eos_token_id=None, pad_token_id=None, bos_token_id=None,
                 settings: Optional[ServingDefaults] = None):
        self.model = model
        cfg = getattr(model, "config", None)
        pick = lambda v, name: v if v is not None else getattr(cfg, name, None)
        self.mask_token_id = pick(mask_token_id, "mask_token_id")
        self.eos_token_id = pick(eos_token_id, "eos_token_id")
        self.pad_token_id = pick(pad_token_id, "pad_token_id")
        self.bos_token_id = pick(bos_token_id, "bos_token_id")
        self.settings = settings or ServingDefaults()

    def generate_ids(self, prompt_ids: Sequence[int], max_tokens: Optional[int] = None, temperature: Optional[float] = None,
                     top_p: Optional[float] = None, top_k: Optional[int] = None, steps: Optional[int] = None,
                     alg: Optional[str] = None, alg_temp: Optional[float] = None, threshold: Optional[float] = None,
                     attention_mask: Optional[torch.Tensor] = None) -> List[int]:
        s = self.settings
        # service.py:188-197: "provided or settings", with `or` for the integer / string fields
        max_tokens = max_tokens or s.max_tokens
        temperature = temperature if temperature is not None else s.temperature
        top_p = top_p if top_p is not None else s.top_p
        top_k = top_k if top_k is not None else s.top_k
        steps = steps or s.steps
        alg = alg or s.alg
        alg_temp = alg_temp if alg_temp is not None else s.alg_temp

        input_ids = torch.as_tensor(list(prompt_ids), dtype=torch.long).unsqueeze(0).to(self.model.device)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)           # the tokenizer's mask for a single unpadded prompt
        prompt_len = input_ids.shape[1]
        gen_config = DLMGenerationConfig(temperature=temperature, top_p=top_p, top_k=top_k, max_new_tokens=max_tokens,
                                         steps=steps, alg=alg, alg_temp=alg_temp, mask_token_id=self.mask_token_id,
                                         eos_token_id=self.eos_token_id, pad_token_id=self.pad_token_id,
                                         bos_token_id=self.bos_token_id, return_dict_in_generate=False)
        extra = {} if threshold is None else dict(threshold=threshold)
        sequences = self.model.diffusion_generate(inputs=input_ids, attention_mask=attention_mask,
                                                  generation_config=gen_config, **extra)
        if hasattr(sequences, "sequences"):
            sequences = sequences.sequences
        # service.py:236-242: tokens after the prompt, mask and pad ids dropped
        out_ids = sequences[0, prompt_len:]
        if self.mask_token_id is not None:
            out_ids = out_ids[out_ids != self.mask_token_id]
        if self.pad_token_id is not None:
            out_ids = out_ids[out_ids != self.pad_token_id]
        return out_ids.tolist()
