"""Host logic of the serving mirror (coda_amd/serving.py) with a stub model: defaults of
serving/fast-api/app/core/config.py:35-42 and the id post-processing of service.py:236-242.  CPU only."""
import types

import torch

from coda_amd.serving import SamplerService, ServingDefaults

MASK, PAD, EOS = 1023, 1000, 1001


class Stub:
    device = torch.device("cpu")
    config = types.SimpleNamespace(mask_token_id=MASK, pad_token_id=PAD, eos_token_id=EOS, bos_token_id=PAD)

    def __init__(self):
        self.calls = []

    def diffusion_generate(self, inputs=None, attention_mask=None, generation_config=None, **kw):
        self.calls.append((inputs, attention_mask, generation_config, kw))
        tail = torch.tensor([[5, MASK, 6, PAD, 7, EOS, MASK, 8]])
        return torch.cat([inputs, tail], dim=1)


def test_defaults_and_postprocessing():
    m = Stub()
    s = SamplerService(m)
    out = s.generate_ids([1, 2, 3])
    assert out == [5, 6, 7, EOS, 8]                       # mask and pad ids dropped, nothing else
    ids, attn, gc, kw = m.calls[0]
    assert ids.tolist() == [[1, 2, 3]] and attn.tolist() == [[1, 1, 1]] and kw == {}
    d = ServingDefaults()
    assert (gc.max_new_tokens, gc.steps, gc.alg, gc.alg_temp, gc.temperature, gc.top_p, gc.top_k) == \
        (d.max_tokens, d.steps, d.alg, d.alg_temp, d.temperature, None, None) == (256, 128, "entropy", 0.1, 0.0, None, None)
    assert gc.return_dict_in_generate is False and gc.mask_token_id == MASK and gc.eos_token_id == EOS


def test_overrides_and_threshold_keyword():
    m = Stub()
    s = SamplerService(m)
    s.generate_ids([9], max_tokens=32, steps=16, alg="confidence_threshold", temperature=0.3, top_p=0.9, alg_temp=0.0, threshold=0.8)
    _, _, gc, kw = m.calls[0]
    assert (gc.max_new_tokens, gc.steps, gc.alg, gc.temperature, gc.top_p, gc.alg_temp) == (32, 16, "confidence_threshold", 0.3, 0.9, 0.0)
    assert kw == {"threshold": 0.8}
    # `steps or settings.steps`: 0 falls back to the default like the reference's `or`
    s.generate_ids([9], steps=0)
    assert m.calls[1][2].steps == 128
