"""Host logic of the id-level lm-eval adapter (coda_amd/lm_eval_adapter.py) against the oracle's restatement of
DiffLLM._generate_batch (evaluation/lm_eval/lm_eval/models/diffllm.py:265-295).  CPU only: the model is a stub."""
import types

import pytest
import torch

from coda_amd.lm_eval_adapter import DiffLLMAdapter, cut_at, left_pad
from oracle import ref_cpu

PAD, EOS, MASK = 900, 901, 1023


class StubModel:
    """Records the keyword set it is called with and returns a canned continuation."""
    device = torch.device("cpu")
    config = types.SimpleNamespace(pad_token_id=PAD, eos_token_id=EOS)

    def __init__(self):
        self.calls = []

    def continuation(self, ids, G):
        g = torch.Generator().manual_seed(int(ids.sum()) % 1000)
        tail = torch.randint(0, 800, (ids.shape[0], G), generator=g)
        tail[0, 5] = EOS                      # row 0 stops early, row 1 never, row 2 immediately
        if ids.shape[0] > 2:
            tail[2, 0] = EOS
        return torch.cat([ids, tail], dim=1)

    def diffusion_generate(self, inputs, **kw):
        self.calls.append((inputs.clone(), kw))
        return types.SimpleNamespace(sequences=self.continuation(inputs, kw["max_new_tokens"]), history=None)


def test_left_pad_and_cut():
    ids = left_pad([[1, 2, 3], [4], []], PAD)
    assert ids.tolist() == [[1, 2, 3], [PAD, PAD, 4], [PAD, PAD, PAD]]
    with pytest.raises(ValueError):
        left_pad([], PAD)
    assert cut_at([1, 2, 3, 4], [3]) == [1, 2]
    assert cut_at([1, 2, 3, 4], [2, 3]) == [1]
    assert cut_at([1, 2, 3, 4], [9]) == [1, 2, 3, 4]
    assert cut_at([EOS, 1], [EOS]) == []
    assert cut_at([1, 2], []) == [1, 2]


def test_defaults_are_the_reference_constructor_defaults():
    a = DiffLLMAdapter(StubModel())
    assert (a.batch_size, a.max_prompt_len, a.max_new_tokens, a.diffusion_steps) == (1, 1024, 128, 32)
    assert (a.temperature, a.top_p, a.top_k, a.alg, a.alg_temp) == (0.1, 0.9, None, "entropy", 0.0)
    assert a.pad_token_id == PAD and a.eos_token_id == EOS and a.rank == 0 and a.world_size == 1
    with pytest.raises(ValueError):
        DiffLLMAdapter(types.SimpleNamespace(config=types.SimpleNamespace(), device="cpu"))


def test_generate_batch_matches_oracle_restatement():
    m = StubModel()
    a = DiffLLMAdapter(m, batch_size=4, max_prompt_len=6, max_new_tokens=12, diffusion_steps=12, temperature=0.1, top_p=0.9)
    prompts = [[5, 6, 7, 8, 9, 10, 11, 12], [3], [20, 21, 22], [30, 31, 32, 33, 34]]
    got = a.generate_batch_ids(prompts)
    ids, kw = m.calls[0]
    # keyword set of diffllm.py:275-287
    assert set(kw) == {"attention_mask", "max_new_tokens", "output_history", "return_dict_in_generate", "steps", "temperature",
                       "top_p", "top_k", "alg", "alg_temp"}
    assert kw["return_dict_in_generate"] is True and kw["output_history"] is False and kw["steps"] == 12
    assert ids.shape == (4, 6) and ids[0].tolist() == [7, 8, 9, 10, 11, 12] and ids[1].tolist() == [PAD] * 5 + [3]
    assert torch.equal(kw["attention_mask"], ids.ne(PAD))
    want = ref_cpu.generate_batch_ids(lambda pids, attn: m.continuation(pids, 12), prompts, pad_id=PAD, eos_id=EOS, max_prompt_len=6)
    assert got == want
    assert len(got[0]) == 5 and len(got[1]) == 12 and got[2] == []


def test_generate_until_batches_in_request_order():
    m = StubModel()
    a = DiffLLMAdapter(m, batch_size=2, max_prompt_len=8, max_new_tokens=8, diffusion_steps=8)
    reqs = [[1, 2], [3, 4, 5], [6], [7, 8, 9, 10], [11]]
    res = a.generate_until_ids(reqs, until=[[EOS]])
    assert len(m.calls) == 3 and [c[0].shape[0] for c in m.calls] == [2, 2, 1]
    assert len(res) == 5
    # `until` cuts like str.split(s)[0]
    full = a.generate_until_ids(reqs[:2])
    stop = full[1][3:5]
    cut = DiffLLMAdapter(m, batch_size=2, max_prompt_len=8, max_new_tokens=8, diffusion_steps=8).generate_until_ids(reqs[:2], until=[stop])
    assert cut[1] == full[1][:3]
