"""CoDAConfig: the model configuration callers load through HF AutoConfig.

Same field names and defaults as the reference (CoDALanguageModel/model_config.py:13-91,
config.json) so an unmodified CoDA model directory loads.  Only the eval-path fields are used
by this package; the training-only ones are kept so that save/load round-trips are lossless."""
from transformers import PretrainedConfig


class CoDAConfig(PretrainedConfig):
    model_type = "CoDA"
    keys_to_ignore_at_inference = ["past_key_values"]

    _DEFAULTS = dict(
        vocab_size=151936, head_dim=128, hidden_act="silu", hidden_size=2048, intermediate_size=6144,
        num_attention_heads=16, num_hidden_layers=28, num_key_value_heads=8, max_position_embeddings=40960,
        initializer_range=0.02, use_cache=True, use_sliding_window=False, rms_norm_eps=1e-6, rope_scaling=None,
        rope_theta=1000000, sliding_window=None, max_window_layers=28, attention_bias=False, attention_dropout=0.0,
        bos_token_id=151643, eos_token_id=151645, pad_token_id=151643, mask_token_id=151669,
        attention_kernel="flash_attention", prefix_probability=0, truncate_probability=0,
        block_masking_probability=[0.25, 0.5, 0.5, 0.75, 0.25], mask_block_sizes=[4, 8, 16, 32],
        sampling_eps=[0.001, 0.25, 0.5, 0.25, 0.001],
    )

    def __init__(self, tie_word_embeddings=True, **kwargs):
        vals = {k: kwargs.pop(k, v) for k, v in self._DEFAULTS.items()}
        if vals["num_key_value_heads"] is None:
            vals["num_key_value_heads"] = vals["num_attention_heads"]
        if not vals["use_sliding_window"]:
            vals["sliding_window"] = None
        token_ids = {k: vals.pop(k) for k in ("bos_token_id", "eos_token_id", "pad_token_id", "mask_token_id")}
        for k, v in vals.items():
            setattr(self, k, list(v) if isinstance(v, list) else v)
        super().__init__(tie_word_embeddings=tie_word_embeddings, **kwargs)
        for k, v in token_ids.items():
            setattr(self, k, v)

    def rope_is_plain(self) -> bool:
        """True when no RoPE scaling is requested (the only mode the reference implements,
        modeling_coda.py:169-172).  transformers 5 may inject {"rope_type": "default", ...}."""
        rs = getattr(self, "rope_scaling", None)
        if not rs:
            return True
        kind = rs.get("rope_type", rs.get("type", "default")) if isinstance(rs, dict) else "default"
        return kind in (None, "default")
