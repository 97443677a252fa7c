"""Checkpoint packaging (coda_amd/checkpoint.py) against the contract of the reference's save_hf_model.py:
sharded / single-file safetensors -> state dict (:51-90), strict load + HF directory (:93-150)."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from coda_amd import CoDAConfig, checkpoint
from oracle import ref_cpu

SMALL = dict(vocab=1024, hidden=256, inter=512, heads=2, kv_heads=1, head_dim=128, layers=2, mask_id=1023, pad_id=1000,
             max_pos=4096)


def _cfg(d):
    return CoDAConfig(vocab_size=d.vocab, hidden_size=d.hidden, intermediate_size=d.inter, num_attention_heads=d.heads,
                      num_key_value_heads=d.kv_heads, head_dim=d.head_dim, num_hidden_layers=d.layers,
                      mask_token_id=d.mask_id, pad_token_id=d.pad_id, bos_token_id=d.pad_id, eos_token_id=d.pad_id + 1,
                      max_position_embeddings=d.max_pos)


def write_raw(tmp, sharded, sd=None, d=None):
    d = d or ref_cpu.Dims(**SMALL)
    sd = sd or {k: v.to(torch.bfloat16) for k, v in ref_cpu.make_state_dict(d, 5, torch.float32, 8.0, 0.1).items()}
    os.makedirs(tmp, exist_ok=True)
    if sharded:
        keys = list(sd)
        parts = [keys[0::3], keys[1::3], keys[2::3]]
        wm = {}
        for i, ks in enumerate(parts):
            fn = f"model-{i + 1:05d}-of-00003.safetensors"
            save_file({k: sd[k].contiguous() for k in ks}, os.path.join(tmp, fn))
            wm.update({k: fn for k in ks})
        json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(tmp, checkpoint.INDEX_NAME), "w"))
    else:
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(tmp, checkpoint.SINGLE_NAME))
    _cfg(d).save_pretrained(tmp)
    return d, sd


@pytest.mark.parametrize("sharded", [False, True])
def test_load_safetensors_both_formats(tmp_path, sharded):
    d, sd = write_raw(str(tmp_path), sharded)
    got = checkpoint.load_safetensors_to_state_dict(str(tmp_path))
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    lazy = checkpoint.SafetensorsState(str(tmp_path))
    assert len(lazy) == len(sd) and set(lazy) == set(sd)
    k = "model.layers.1.mlp.down_proj.weight"
    assert torch.equal(lazy[k], sd[k]) and lazy.shape_dtype(k)[0] == tuple(sd[k].shape)


def test_missing_files_error_text(tmp_path):
    with pytest.raises(FileNotFoundError, match="No safetensors found in .* Expected 'model.safetensors' or 'model.safetensors.index.json'"):
        checkpoint.load_safetensors_to_state_dict(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        checkpoint.SafetensorsState(str(tmp_path))


def test_expected_shapes_match_the_module():
    from coda_amd import CoDALanguageModel
    d = ref_cpu.Dims(**SMALL)
    cfg = _cfg(d)
    with torch.device("meta"):
        m = CoDALanguageModel(cfg)
    want = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert checkpoint.expected_shapes(cfg) == want
    assert len(want) == 2 + 11 * d.layers + 1


@pytest.mark.parametrize("sharded_out", [False, True])
def test_package_checkpoint_roundtrip(tmp_path, sharded_out):
    import transformers
    raw, out = str(tmp_path / "raw"), str(tmp_path / "hf")
    d, sd = write_raw(raw, sharded=True)
    open(os.path.join(raw, "tokenizer_config.json"), "w").write("{}")
    checkpoint.package_checkpoint(raw, out, max_shard_bytes=(1 << 20) if sharded_out else (1 << 40))
    files = set(os.listdir(out))
    assert {"config.json", "model_config.py", "modeling_coda.py", "tokenizer_config.json"} <= files
    assert (checkpoint.INDEX_NAME in files) == sharded_out and (checkpoint.SINGLE_NAME in files) != sharded_out
    cj = json.load(open(os.path.join(out, "config.json")))
    assert cj["auto_map"] == {"AutoConfig": "model_config.CoDAConfig", "AutoModel": "modeling_coda.CoDALanguageModel"}
    back = checkpoint.load_safetensors_to_state_dict(out)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    # the callers' load path (diffllm.py:207-215, service.py:61-74)
    m = transformers.AutoModel.from_pretrained(out, trust_remote_code=True, torch_dtype=torch.bfloat16)
    got = m.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)


def test_strict_state_check(tmp_path):
    d = ref_cpu.Dims(**SMALL)
    sd = {k: v.to(torch.bfloat16) for k, v in ref_cpu.make_state_dict(d, 5, torch.float32, 8.0, 0.1).items()}
    broken = dict(sd)
    broken.pop("model.norm.weight")
    broken["extra.weight"] = torch.zeros(2)
    broken["lm_head.weight"] = torch.zeros(3, 3)
    write_raw(str(tmp_path / "a"), False, sd=broken, d=d)
    with pytest.raises(RuntimeError) as e:
        checkpoint.package_checkpoint(str(tmp_path / "a"), str(tmp_path / "b"))
    msg = str(e.value)
    assert "Missing key(s) in state_dict" in msg and '"model.norm.weight"' in msg
    assert "Unexpected key(s) in state_dict" in msg and '"extra.weight"' in msg and "size mismatch for lm_head.weight" in msg
