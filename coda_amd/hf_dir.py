"""Writes an HF remote-code model directory with the reference's layout (config.json auto_map ->
model_config.CoDAConfig / modeling_coda.CoDALanguageModel, CoDALanguageModel/config.json:8-11), so callers keep
using ``AutoModel.from_pretrained(dir, trust_remote_code=True)`` and get the MI355X implementation."""
import json
import os

_SHIM = '''"""Remote-code shim: resolves to the MI355X-native implementation in the installed coda_amd package."""
from coda_amd.{module} import *  # noqa: F401,F403
from coda_amd.{module} import {symbol}  # noqa: F401
'''


def write_code_shims(path: str) -> None:
    with open(os.path.join(path, "model_config.py"), "w") as f:
        f.write(_SHIM.format(module="model_config", symbol="CoDAConfig"))
    with open(os.path.join(path, "modeling_coda.py"), "w") as f:
        f.write(_SHIM.format(module="modeling_coda", symbol="CoDALanguageModel"))


def patch_config_json(path: str) -> None:
    p = os.path.join(path, "config.json")
    cfg = json.load(open(p))
    cfg["architectures"] = ["CoDALanguageModel"]
    cfg["auto_map"] = {"AutoConfig": "model_config.CoDAConfig", "AutoModel": "modeling_coda.CoDALanguageModel"}
    cfg["model_type"] = "CoDA"
    json.dump(cfg, open(p, "w"), indent=2)


def save_model_dir(model, path: str) -> None:
    """save_pretrained + the code shims + auto_map.  An existing CoDA checkpoint directory can be
    converted in place with write_code_shims(dir) alone (its config.json already has the auto_map)."""
    os.makedirs(path, exist_ok=True)
    model.save_pretrained(path)
    write_code_shims(path)
    patch_config_json(path)
