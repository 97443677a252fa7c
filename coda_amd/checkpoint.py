"""Checkpoint packaging for the sampler path.

Reference: ``save_hf_model.py`` - ``load_safetensors_to_state_dict`` (:51-90, sharded or single-file safetensors ->
state dict) and the packaging script (:93-150: instantiate ``CoDALanguageModel(config)``, strict ``load_state_dict``,
``save_pretrained`` + tokenizer into an HF remote-code directory).

Here:
  load_safetensors_to_state_dict   same contract (same file names, same FileNotFoundError)
  package_checkpoint               raw safetensors + config.json -> HF directory (weights re-sharded by safetensors,
                                   config.json with the reference's auto_map, code shims resolving to coda_amd,
                                   tokenizer files copied when present); the strict key/shape check of
                                   ``load_state_dict`` is done against the shapes the config implies, without
                                   building a second copy of the 1.7 B parameters
  SafetensorsState                 lazy Mapping key -> tensor over a (sharded) checkpoint
  load_packed                      checkpoint directory -> model whose weights exist ONLY in the engine's fused HBM
                                   layouts (q|k|v concatenated and row-permuted for the fused RoPE epilogue, gate|up
                                   interleaved): every tensor goes disk -> GPU -> packed buffer once; the nn.Module
                                   tree stays on the meta device
"""
from __future__ import annotations

import glob
import json
import os
import shutil
from collections.abc import Mapping
from typing import Dict, Optional, Tuple

import torch

from .hf_dir import patch_config_json, write_code_shims
from .model_config import CoDAConfig

INDEX_NAME = "model.safetensors.index.json"
SINGLE_NAME = "model.safetensors"
_TOKENIZER_GLOBS = ("tokenizer*", "vocab.json", "merges.txt", "special_tokens_map.json", "added_tokens.json", "chat_template*")


def _not_found(model_dir):
    return FileNotFoundError(
        f"No safetensors found in {model_dir}. Expected 'model.safetensors' or 'model.safetensors.index.json'.")


def load_safetensors_to_state_dict(model_dir: str) -> Dict[str, torch.Tensor]:
    """save_hf_model.py:51-90.  Sharded (index json) wins over single-file when both exist, as in the reference."""
    from safetensors import safe_open
    from safetensors.torch import load_file
    index_file, single_file = os.path.join(model_dir, INDEX_NAME), os.path.join(model_dir, SINGLE_NAME)
    state: Dict[str, torch.Tensor] = {}
    if os.path.exists(index_file):
        with open(index_file) as f:
            weight_map = json.load(f)["weight_map"]
        for filename in sorted(set(weight_map.values())):
            with safe_open(os.path.join(model_dir, filename), framework="pt", device="cpu") as f:
                for key in f.keys():
                    state[key] = f.get_tensor(key)
    elif os.path.exists(single_file):
        state = load_file(single_file)
    else:
        raise _not_found(model_dir)
    return state


class SafetensorsState(Mapping):
    """key -> tensor, read from disk on access (optionally straight onto ``device``); nothing is cached."""

    def __init__(self, model_dir: str, device: str = "cpu"):
        from safetensors import safe_open
        self._open, self._dir, self._device = safe_open, model_dir, str(device)
        index_file, single_file = os.path.join(model_dir, INDEX_NAME), os.path.join(model_dir, SINGLE_NAME)
        if os.path.exists(index_file):
            with open(index_file) as f:
                self._where = dict(json.load(f)["weight_map"])
        elif os.path.exists(single_file):
            with safe_open(single_file, framework="pt", device="cpu") as f:
                self._where = {k: SINGLE_NAME for k in f.keys()}
        else:
            raise _not_found(model_dir)

    def __getitem__(self, key):
        with self._open(os.path.join(self._dir, self._where[key]), framework="pt", device=self._device) as f:
            return f.get_tensor(key)

    def shape_dtype(self, key) -> Tuple[Tuple[int, ...], str]:
        with self._open(os.path.join(self._dir, self._where[key]), framework="pt", device="cpu") as f:
            sl = f.get_slice(key)
            return tuple(sl.get_shape()), sl.get_dtype()

    def __iter__(self):
        return iter(self._where)

    def __len__(self):
        return len(self._where)


def expected_shapes(cfg: CoDAConfig) -> Dict[str, Tuple[int, ...]]:
    """State-dict keys and shapes of ``CoDALanguageModel(cfg)`` (2 + 11 per layer + 1 tensors)."""
    H, I, V, d = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.head_dim
    q, kv = cfg.num_attention_heads * d, cfg.num_key_value_heads * d
    out = {"model.embed_tokens.weight": (V, H), "model.norm.weight": (H,), "lm_head.weight": (V, H)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        out.update({p + "self_attn.q_proj.weight": (q, H), p + "self_attn.k_proj.weight": (kv, H),
                    p + "self_attn.v_proj.weight": (kv, H), p + "self_attn.o_proj.weight": (H, q),
                    p + "self_attn.q_norm.weight": (d,), p + "self_attn.k_norm.weight": (d,),
                    p + "mlp.gate_proj.weight": (I, H), p + "mlp.up_proj.weight": (I, H), p + "mlp.down_proj.weight": (H, I),
                    p + "input_layernorm.weight": (H,), p + "post_attention_layernorm.weight": (H,)})
    return out


def check_state(cfg: CoDAConfig, shapes: Dict[str, Tuple[int, ...]]) -> None:
    """The strict ``load_state_dict`` of save_hf_model.py:142 as a shape check: same RuntimeError text structure."""
    want = expected_shapes(cfg)
    missing = [k for k in want if k not in shapes]
    unexpected = [k for k in shapes if k not in want]
    bad = [f"size mismatch for {k}: copying a param with shape {tuple(shapes[k])} from checkpoint, the shape in current "
           f"model is {want[k]}." for k in want if k in shapes and tuple(shapes[k]) != want[k]]
    msgs = []
    if unexpected:
        msgs.append("Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
    if missing:
        msgs.append("Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
    msgs += bad
    if msgs:
        raise RuntimeError("Error(s) in loading state_dict for CoDALanguageModel:\n\t" + "\n\t".join(msgs))


def read_config(path: str) -> CoDAConfig:
    with open(path) as f:
        d = json.load(f)
    for k in ("architectures", "auto_map", "model_type", "transformers_version"):
        d.pop(k, None)
    return CoDAConfig(**d)


def package_checkpoint(checkpoint_dir: str, output_dir: str, config_path: Optional[str] = None, model_dir: Optional[str] = None,
                       max_shard_bytes: int = 5 * 2 ** 30) -> str:
    """save_hf_model.py:93-150 without the module round trip.  ``config_path`` defaults to ``checkpoint_dir/config.json``
    (the reference reads ``<model_name>/config.json``); ``model_dir`` is where tokenizer files are copied from."""
    from safetensors.torch import save_file
    config_path = config_path or os.path.join(checkpoint_dir, "config.json")
    cfg = read_config(config_path)
    src = SafetensorsState(checkpoint_dir)
    check_state(cfg, {k: src.shape_dtype(k)[0] for k in src})
    os.makedirs(output_dir, exist_ok=True)
    # shard in state-dict order, one tensor resident at a time beyond the shard being written
    order = list(expected_shapes(cfg))
    shards, cur, cur_bytes = [], {}, 0
    for k in order:
        t = src[k].contiguous()
        nbytes = t.numel() * t.element_size()
        if cur and cur_bytes + nbytes > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = t
        cur_bytes += nbytes
    shards.append(cur)
    for name in (SINGLE_NAME, INDEX_NAME):
        if os.path.exists(os.path.join(output_dir, name)):
            os.remove(os.path.join(output_dir, name))
    if len(shards) == 1:
        save_file(shards[0], os.path.join(output_dir, SINGLE_NAME), metadata={"format": "pt"})
    else:
        weight_map, total = {}, 0
        for i, sh in enumerate(shards):
            fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(output_dir, fn), metadata={"format": "pt"})
            for k, t in sh.items():
                weight_map[k] = fn
                total += t.numel() * t.element_size()
        with open(os.path.join(output_dir, INDEX_NAME), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
    shutil.copyfile(config_path, os.path.join(output_dir, "config.json"))
    patch_config_json(output_dir)
    write_code_shims(output_dir)
    for pat in _TOKENIZER_GLOBS:
        for fn in glob.glob(os.path.join(model_dir or checkpoint_dir, pat)):
            if os.path.isfile(fn):
                shutil.copy(fn, output_dir)
    return output_dir


def load_packed(model_dir: str, device="cuda", config: Optional[CoDAConfig] = None):
    """Checkpoint directory -> ``CoDALanguageModel`` whose weights live only in the HIP engine's fused layouts."""
    from .engine import HipEngine
    from .modeling_coda import CoDALanguageModel
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    cfg = config or read_config(os.path.join(model_dir, "config.json"))
    state = SafetensorsState(model_dir, device=str(device))
    check_state(cfg, {k: state.shape_dtype(k)[0] for k in state})
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("meta"):
            model = CoDALanguageModel(cfg)
    finally:
        torch.set_default_dtype(old)
    model.eval()
    model.pin_engine(HipEngine(cfg, state, device))
    return model
