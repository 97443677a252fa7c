"""CoDALanguageModel on MI355X: the HF-loadable model class callers get from AutoModel.

State-dict keys, constructor, ``forward(input_ids) -> (logits, None)`` in eval mode,
``diffusion_generate`` and the attributes callers touch (.device, .config, .generation_config,
.mask_token_id, .get_embeds) mirror the reference (CoDALanguageModel/modeling_coda.py:258-412).
The nn.Module tree below only HOLDS parameters (so checkpoints load unchanged, lm_head untied:
SURVEY.md F4); the arithmetic of the eval path is done by the HIP engine (coda_amd/engine.py).
The training branch of the reference's forward (modeling_coda.py:414-558) is out of scope.
"""
from __future__ import annotations

import torch
from torch import nn
from transformers import PreTrainedModel

from . import _lib
from .generation_utils import DLMGenerationConfig, DLMGenerationMixin
from .model_config import CoDAConfig


class CoDARMSNorm(nn.Module):
    """Parameter holder for an RMSNorm weight (modeling_coda.py:30-41)."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps


class CoDAAttention(nn.Module):
    def __init__(self, config: CoDAConfig, layer_idx=None):
        super().__init__()
        d, H = config.head_dim, config.hidden_size
        bias = getattr(config, "attention_bias", False)
        if bias:
            raise NotImplementedError("attention_bias=True is not supported by the HIP engine")
        self.layer_idx = layer_idx
        self.q_proj = nn.Linear(H, config.num_attention_heads * d, bias=False)
        self.k_proj = nn.Linear(H, config.num_key_value_heads * d, bias=False)
        self.v_proj = nn.Linear(H, config.num_key_value_heads * d, bias=False)
        self.o_proj = nn.Linear(config.num_attention_heads * d, H, bias=False)
        self.q_norm = CoDARMSNorm(d, eps=config.rms_norm_eps)
        self.k_norm = CoDARMSNorm(d, eps=config.rms_norm_eps)


class CoDAMLP(nn.Module):
    def __init__(self, config: CoDAConfig):
        super().__init__()
        if config.hidden_act != "silu":
            raise NotImplementedError("only hidden_act='silu' is implemented (SwiGLU epilogue)")
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)


class CoDADecoderLayer(nn.Module):
    def __init__(self, config: CoDAConfig, layer_idx: int):
        super().__init__()
        self.self_attn = CoDAAttention(config, layer_idx)
        self.mlp = CoDAMLP(config)
        self.input_layernorm = CoDARMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = CoDARMSNorm(config.hidden_size, eps=config.rms_norm_eps)


def _init_linear_embedding(module, std):
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=std)
    elif isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=std)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()


class CoDAModel(PreTrainedModel):
    config_class = CoDAConfig

    def __init__(self, config: CoDAConfig):
        super().__init__(config)
        if not config.rope_is_plain():
            raise NotImplementedError("Scaling is not implemented")
        pad = getattr(config, "pad_token_id", None)
        self.padding_idx = pad if (pad is not None and pad < config.vocab_size) else None
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=self.padding_idx)
        self.layers = nn.Sequential(*[CoDADecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = CoDARMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_init()

    def _init_weights(self, module):
        _init_linear_embedding(module, getattr(self.config, "initializer_range", 0.02))


class CoDALanguageModel(DLMGenerationMixin, PreTrainedModel):
    config_class = CoDAConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = False
    _no_split_modules = ["CoDADecoderLayer"]

    def __init__(self, config: CoDAConfig):
        super().__init__(config)
        self.config = config
        self.model = CoDAModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.mask_token_id = config.mask_token_id
        self.generation_config = DLMGenerationConfig(mask_token_id=config.mask_token_id)
        self._engine = None
        self._engine_key = None
        self.post_init()

    def _init_weights(self, module):
        _init_linear_embedding(module, getattr(self.config, "initializer_range", 0.02))

    def get_embeds(self, input_ids):
        return self.model.embed_tokens(input_ids)

    # ------------------------------------------------------------------ HIP engine plumbing
    def hip_engine(self):
        """The packed-weight engine for the device the parameters live on (built lazily; rebuilt when
        parameters were replaced or modified in place).  Raises if libcoda_hip.so is absent or
        the model is not on a GPU: there is no PyTorch fallback."""
        from .engine import HipEngine
        if getattr(self, "_engine_pinned", False):
            return self._engine
        w = self.lm_head.weight
        key = (w.device, w.data_ptr(), w._version, self.model.embed_tokens.weight.data_ptr(),
               self.model.embed_tokens.weight._version)
        if self._engine is None or self._engine_key != key:
            if w.dtype != torch.bfloat16:
                import warnings
                warnings.warn(f"CoDA HIP engine computes in bf16; {w.dtype} parameters are packed to bf16.", UserWarning)
            self._engine = HipEngine(self.config, self.state_dict(), w.device)
            self._engine_key = key
        return self._engine

    def pin_engine(self, engine):
        """Adopts an engine whose packed weights are the ONLY copy (coda_amd.checkpoint.load_packed: the module tree is
        on the meta device); ``device`` then reports the engine's GPU."""
        self._engine, self._engine_pinned = engine, True

    @property
    def device(self):
        if getattr(self, "_engine_pinned", False):
            return self._engine.device
        return super().device

    def refresh_engine(self):
        """Drops the packed weights (call after editing parameters other than lm_head / embeddings in place)."""
        if getattr(self, "_engine_pinned", False):
            raise _lib.CodaHipError("this model's weights exist only in its packed engine (load_packed); reload the checkpoint instead")
        self._engine = None

    def forward(self, input_ids: torch.LongTensor, labels=None, attention_mask=None, src_mask=None,
                training_mode: str = "pretrain", **kwargs):
        """Eval branch of the reference forward (modeling_coda.py:406-412): (logits [B,L,V], None).
        ``attention_mask``/``labels``/``src_mask`` are ignored there and here."""
        if self.training:
            raise NotImplementedError("coda_amd implements the inference path only; call model.eval()")
        if input_ids.device.type != "cuda":
            raise _lib.CodaHipError("CoDALanguageModel.forward runs on an MI355X only; move the model and inputs to cuda")
        logits = self.hip_engine().logits(input_ids)
        w_dtype = self.lm_head.weight.dtype
        return (logits if logits.dtype == w_dtype else logits.to(w_dtype)), None
