"""CPU-side checks: C-ABI library exports, config/generation-config contract, schedules, DP sharding."""
import math
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generation_config_fields_and_defaults():
    from coda_amd import DLMGenerationConfig
    gc = DLMGenerationConfig()
    expect = dict(temperature=0.0, top_p=None, top_k=None, max_length=20, max_new_tokens=None, eps=1e-3, steps=512,
                  alg="origin", alg_temp=None, num_return_sequences=1, return_dict_in_generate=False,
                  output_history=False, mask_token_id=None, pad_token_id=None, bos_token_id=None, eos_token_id=None)
    for k, v in expect.items():
        assert getattr(gc, k) == v, k
    assert gc.block_length is None
    left = gc.update(steps=7, alg="entropy", not_a_field=3)
    assert gc.steps == 7 and gc.alg == "entropy" and left == {"not_a_field": 3}
    assert gc.validate() is None


def test_config_defaults_match_reference_config_json():
    import json
    from coda_amd import CoDAConfig
    c = CoDAConfig()
    ref = dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_attention_heads=16,
               num_key_value_heads=8, head_dim=128, num_hidden_layers=28, max_position_embeddings=40960,
               rms_norm_eps=1e-6, rope_theta=1000000, mask_token_id=151669, pad_token_id=151643, bos_token_id=151643,
               eos_token_id=151645, attention_kernel="flash_attention", hidden_act="silu", model_type="CoDA")
    for k, v in ref.items():
        assert getattr(c, k) == v, k
    c2 = CoDAConfig.from_dict(json.loads(c.to_json_string()))
    assert c2.mask_token_id == 151669 and c2.rope_is_plain()


def _tiny_model():
    from coda_amd import CoDAConfig, CoDALanguageModel
    cfg = CoDAConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_attention_heads=2,
                     num_key_value_heads=1, head_dim=128, num_hidden_layers=2, mask_token_id=1023, pad_token_id=1000,
                     bos_token_id=1000, eos_token_id=1001)
    return cfg, CoDALanguageModel(cfg).eval()


def test_state_dict_keys_and_untied_lm_head():
    from oracle import ref_cpu
    cfg, m = _tiny_model()
    d = ref_cpu.Dims(vocab=1024, hidden=256, inter=512, heads=2, kv_heads=1, head_dim=128, layers=2, mask_id=1023)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == d.shapes()
    assert m.lm_head.weight.data_ptr() != m.model.embed_tokens.weight.data_ptr()
    assert m.mask_token_id == 1023 and m.generation_config.mask_token_id == 1023
    assert m.get_embeds(torch.tensor([[1, 2]])).shape == (1, 2, 256)


def test_auto_model_directory_roundtrip(tmp_path):
    """HF remote-code directory (config.json auto_map + model_config.py / modeling_coda.py), as callers load it
    (evaluation/lm_eval/lm_eval/models/diffllm.py:207-215)."""
    from coda_amd.hf_dir import save_model_dir
    from transformers import AutoModel, AutoConfig
    cfg, m = _tiny_model()
    save_model_dir(m, str(tmp_path))
    assert {"config.json", "model_config.py", "modeling_coda.py", "model.safetensors"} <= set(os.listdir(tmp_path))
    c = AutoConfig.from_pretrained(str(tmp_path), trust_remote_code=True)
    assert type(c).__name__ == "CoDAConfig" and c.mask_token_id == 1023
    m2 = AutoModel.from_pretrained(str(tmp_path), trust_remote_code=True, torch_dtype=torch.bfloat16)
    assert type(m2).__name__ == "CoDALanguageModel" and hasattr(m2, "diffusion_generate")
    assert m2.lm_head.weight.dtype == torch.bfloat16
    assert torch.equal(m2.lm_head.weight.float(), m.lm_head.weight.to(torch.bfloat16).float())


def test_no_cpu_fallback_and_argument_errors():
    from coda_amd._lib import CodaHipError
    cfg, m = _tiny_model()
    ids = torch.zeros(1, 4, dtype=torch.long)
    with pytest.raises(CodaHipError):
        m(ids)
    with pytest.raises(CodaHipError):
        m.diffusion_generate(ids, max_new_tokens=8, steps=4, alg="entropy")
    with pytest.raises(ValueError):
        m.diffusion_generate(torch.zeros(1, 40, dtype=torch.long), max_length=20, steps=4, alg="entropy")
    with pytest.raises(AssertionError):
        m.diffusion_generate(None, max_new_tokens=8)
    with pytest.raises(RuntimeError, match="Unknown alg"):
        m.diffusion_generate(ids, max_new_tokens=8, steps=4, alg="bogus")
    m.train()
    with pytest.raises(NotImplementedError):
        m(ids)


def test_non_device_tensors_raise_before_any_launch():
    """A model on the GPU handed CPU tensors: the reference raises a Python device-mismatch error inside embed_tokens;
    here the tensor's address would reach a kernel, so it must be a Python exception before any launch (checked without
    a GPU: a stand-in engine that reports a cuda device and fails the test if anything tries to run on it)."""
    import types
    import warnings
    from coda_amd._lib import CodaHipError
    from coda_amd.engine import HipEngine
    cfg, m = _tiny_model()

    class NoLaunch:
        device = torch.device("cuda", 0)

        def __getattr__(self, name):
            raise AssertionError(f"engine.{name} reached with a non-device tensor")
    m.pin_engine(NoLaunch())
    assert m.device == torch.device("cuda", 0)
    ids = torch.zeros(1, 4, dtype=torch.long)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with pytest.raises(CodaHipError, match="input_ids is on cpu"):
            m.diffusion_generate(ids, max_new_tokens=8, steps=4, alg="entropy")
    assert any("different than your model's device" in str(x.message) for x in w)      # the reference's warning stays
    with pytest.raises(CodaHipError):
        m._sample(ids, None, m._prepare_generation_config(None, max_length=12, steps=4, alg="entropy"), None, None)
    # engine-level guard (ids, injected noise, user buffers): host tensors, wrong dtypes
    fake = types.SimpleNamespace(device=torch.device("cuda", 0))
    with pytest.raises(CodaHipError, match="selection noise is on cpu"):
        HipEngine._dev(fake, torch.ones(2, 3), "selection noise", torch.float32)
    with pytest.raises(CodaHipError, match="is on list"):
        HipEngine._dev(fake, [1, 2], "x")
    assert HipEngine._dev(fake, None, "noise") is None
    with pytest.raises(CodaHipError):
        HipEngine.forward_hidden(fake, ids)


def test_ids_outside_vocabulary_raise_like_nn_embedding():
    cfg, m = _tiny_model()
    for bad in (torch.tensor([[1, 2, 5000]]), torch.tensor([[-1, 2, 3]])):
        with pytest.raises(IndexError, match="index out of range"):
            m.diffusion_generate(bad, max_new_tokens=8, steps=4, alg="entropy")
    with pytest.raises(IndexError, match="mask_token_id"):
        m.diffusion_generate(torch.tensor([[1, 2, 3]]), max_new_tokens=8, steps=4, alg="entropy", mask_token_id=4096)


def test_pad_defaults_to_eos_with_warning(caplog):
    """generation_utils.py:316-325: pad_token_id unset + eos set -> pad := eos[0], with the reference's log line."""
    import logging
    from coda_amd import DLMGenerationConfig
    from coda_amd import generation_utils as gu
    cfg, m = _tiny_model()
    for eos, want in ((1001, 1001), ([1001, 7], 1001), (torch.tensor([5, 6]), 5)):
        gc = DLMGenerationConfig(eos_token_id=eos, mask_token_id=1023)
        gu.logger.propagate = True
        with caplog.at_level(logging.WARNING):
            caplog.clear()
            m._prepare_special_tokens(gc)
        assert gc.pad_token_id == want
        assert any("Setting `pad_token_id` to `eos_token_id`" in r.getMessage() for r in caplog.records)
    gc = DLMGenerationConfig(eos_token_id=1001, pad_token_id=3)
    m._prepare_special_tokens(gc)
    assert gc.pad_token_id == 3
    gc = DLMGenerationConfig()
    m._prepare_special_tokens(gc)
    assert gc.pad_token_id is None


def test_per_call_state_is_not_kept_on_the_model():
    cfg, m = _tiny_model()
    from coda_amd._lib import CodaHipError
    with pytest.raises(CodaHipError):
        m.diffusion_generate(torch.zeros(1, 4, dtype=torch.long), max_new_tokens=8, steps=4, alg="entropy", threshold=0.3,
                             debug_trace=[])
    assert not any(hasattr(m, a) for a in ("_threshold", "_debug_trace", "_prompt_has_mask", "_prompt_has_pad"))


def test_window_plan_and_schedule_match_oracle():
    from coda_amd.generation_utils import plan_windows, fp32_schedule
    from oracle import ref_cpu
    assert plan_windows(128, 256, 128, 32) == (8, 32, 16)
    assert plan_windows(128, 256, 128, None) == (1, 256, 128)
    assert plan_windows(128, 256, 128, 256) == (1, 256, 128)
    with pytest.raises(ValueError):
        plan_windows(8, 100, 10, 32)
    with pytest.raises(ValueError):
        plan_windows(8, 64, 9, 32)
    for G, steps in ((256, 128), (512, 32), (32, 16), (64, 32), (768, 768)):
        ratios = fp32_schedule(steps, 1e-3)
        m, ks = torch.tensor(G * 4), []
        for j in range(steps):
            nm = m / 4
            k = int(nm) if j == steps - 1 else int(nm * torch.tensor(ratios[j]))
            ks.append(k)
            m = m - 4 * k
        assert ks == ref_cpu.transfer_counts(G * 4, 4, steps, 1e-3)


def test_shard_bounds_cover_everything():
    from coda_amd.dp import shard_bounds
    for n in (0, 1, 7, 32, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_data_parallel_gather_gloo_world2(tmp_path):
    """N>1 path on CPU: two gloo ranks shard 5 prompts (ragged 3+2), 'generate' with a stub, gather in order."""
    script = tmp_path / "dp_worker.py"
    script.write_text(f"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {ROOT!r})
from coda_amd.dp import generate_data_parallel, gather_completions
dist.init_process_group("gloo")
ids = torch.arange(5 * 4).reshape(5, 4)
gen = lambda x: torch.cat([x, x.sum(1, keepdim=True) + 100 * dist.get_rank()], dim=1)
out = generate_data_parallel(gen, ids, dist)
expect = torch.cat([ids, ids.sum(1, keepdim=True) + torch.tensor([0, 0, 0, 100, 100])[:, None]], dim=1)
assert torch.equal(out, expect), out
m = gather_completions(torch.full((1, 2), dist.get_rank()), dist)
assert m.tolist() == [[0, 0], [1, 1]]
dist.destroy_process_group()
print("rank ok")
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("rank ok") == 2


def test_bench_spawns_its_own_ranks_and_gathers_metrics():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks itself (gloo on the CPU around a stub
    generate here; RCCL, one rank per GPU, on a node) and rank 0 prints ONE line whose n_gpus is the process group's
    world size, with the per-rank metrics vector gathered (evaluator.py:523-526)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-cpu",
                        "--batch", "3", "--prompt", "4", "--new-tokens", "8", "--denoise-steps", "4", "--block-length", "4"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 6 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["valid"] is False and "STUB" in line["data"]
    pr = line["per_rank"]
    assert pr["generated_tokens"] == [3 * 8 * 2.0] * 2 and pr["rows_after_gather"] == [6.0, 6.0]
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 2 - 2 * 3 * 8 * 2) < 1e-3      # value = all ranks' tokens / time
    # a rank that fails takes the run down with a non-zero status instead of hanging the others
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-cpu",
                        "--new-tokens", "8", "--denoise-steps", "4", "--block-length", "4"],
                       env=dict(env, CODA_BENCH_TEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "stub rank asked to fail" in r.stderr


def test_three_operation_division_is_correctly_rounded():
    """coda_div_by (csrc/coda_common.h): q0 = RN(x rt), r = fma(-q0, t, x), q1 = fma(r, rt, q0) with rt = RN(1 / t) must equal
    the IEEE fp32 quotient x / t (the reference's `logits / temperature`) for EVERY bf16 x; checked in exact rational
    arithmetic (each fp32 operation = one rounding of an exact value)."""
    from fractions import Fraction
    import numpy as np

    def rn32(fr):                       # exact Fraction -> nearest float32 (ties to even), as a Fraction
        if fr == 0:
            return Fraction(0)
        sign = -1 if fr < 0 else 1
        a = abs(fr)
        e = a.numerator.bit_length() - a.denominator.bit_length()
        if Fraction(2) ** e > a:
            e -= 1
        e = max(e, -126)
        scaled = a / Fraction(2) ** (e - 23)          # in [2^23, 2^24) for normal numbers
        n = scaled.numerator // scaled.denominator
        rem = scaled - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
            n += 1
        return sign * n * Fraction(2) ** (e - 23)
    bits = np.arange(65536, dtype=np.uint32)
    xs = (bits << 16).view(np.float32)
    xs = xs[np.isfinite(xs) & (np.abs(xs) < 1e30) & ((np.abs(xs) > 1e-30) | (xs == 0))]     # where x / t stays normal
    xs = xs[::7]                                      # every 7th bf16 value (odd stride: all mantissa patterns), ~9000 points
    for t in (0.1, 0.7, 0.5, 0.8, 1.3, 0.6):
        t32 = Fraction(float(np.float32(t)))
        rt = rn32(1 / t32)
        assert float(rt) == float(np.float32(1.0 / float(np.float32(t))))                     # the host's (float)(1.0 / (double)t)
        bad = 0
        for x in xs:
            xf = Fraction(float(x))
            q0 = rn32(xf * rt)
            r = rn32(xf - q0 * t32)
            q1 = rn32(q0 + r * rt)
            bad += q1 != rn32(xf / t32)
        assert bad == 0, (t, bad)


def test_hash_uniform_stays_strictly_inside_the_unit_interval():
    """coda_hash_uniform's mapping of the 32-bit hash to fp32 (restated in tests/parity.py): every value exact in fp32 and
    strictly inside (0, 1).  With 24 bits + 0.5 the top value rounded to 1.0, -log(u) became 0 and that entry's race key
    +inf: it won the exponential race whatever its logit (found by the cross-check of the two lm-head kernels)."""
    import numpy as np
    top = np.float32(np.float32(0xFFFFFFFF >> 9) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    bot = np.float32(np.float32(0) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    assert 0.0 < float(bot) == 2.0 ** -24 and float(top) == 1.0 - 2.0 ** -24 < 1.0
    noise = [-math.log(-math.log(float(u))) for u in (bot, top)]
    assert -2.82 < noise[0] < noise[1] < 16.64          # the bounds CODA_RACE_WINDOW (20.5) is built on
    old_top = np.float32(np.float32(0xFFFFFFFF >> 8) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    assert float(old_top) == 1.0                         # what the previous mapping did
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity import hash_uniform_host
    u = hash_uniform_host(0x1234567890, torch.arange(64)[:, None], torch.arange(4096)[None, :])
    assert float(u.min()) >= 2.0 ** -24 and float(u.max()) <= 1.0 - 2.0 ** -24


def test_fast_division_is_licensed_per_temperature_by_the_library():
    """coda_div_by's three-operation quotient is used only for temperatures at which the library has verified it, on the
    host, against the IEEE quotient for EVERY bf16 input of its fast-path domain (csrc/abi_misc.hip); anything else - and
    every fp32-logits call - takes the true division (the ABI accepts any temperature).  Host-only call."""
    import numpy as np
    from coda_amd import _lib
    lib = _lib.load()
    for t in (0.1, 0.7, 0.5, 0.8, 1.3, 0.6, 1.0):          # the configurations' temperatures run on the fast form
        assert lib.coda_temperature_fast_division(t) == 1, t
    for t in (0.0, -0.5, 1e-6, 5e4, float("nan")):         # outside its stated range: division
        assert lib.coda_temperature_fast_division(t) == 0, t
    rng = np.random.default_rng(5)
    flags = [lib.coda_temperature_fast_division(float(np.float32(t))) for t in rng.uniform(0.02, 4.0, 300)]
    assert set(flags) <= {0, 1}                              # every verdict is a verified one, whichever way it falls
    print(f"fast division licensed at {sum(flags)} of {len(flags)} random temperatures")
