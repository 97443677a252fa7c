"""Host side of the MI355X sampler engine: packed weights, HBM workspaces and the kernel launch
sequence of one decoder forward / one denoise step.  PyTorch is used for device memory and
streams only; all arithmetic is done by libcoda_hip.so (see include/coda_hip.h).

HBM layout (bf16 unless noted; T = B*L tokens, row-major, all buffers resident for the whole
generation - nothing is allocated inside the denoise loop and no torch kernel runs in it: sampling
noise of every kind comes from an in-kernel counter hash unless a test injects it):
  weights   embed [V,H]; per layer  wqkv [(n_q+2 n_kv)*128, H] (q|k|v rows concatenated),
            wo [H, n_q*128], wgu [2I, H] (gate/up interleaved in blocks of 16 rows so that one
            MFMA accumulator pair holds gate and up of the same feature), wd [H, I], norm vectors;
            final norm [H]; lm_head [V,H]
  state     h [T,H] residual stream, a [T,H] normed input, qkv [T,(n_q+2 n_kv)*128],
            ao [T,n_q*128] attention output, mlp [T,I], hn [T,H] final-normed hidden
  sampler   src_rows/dst_pos int32 [B*W], count int32[1], conf fp32 [T], x0 int64 [T],
            lm-head partial-statistics workspace (<= 4.2 MB), k int32[1]
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _on_device(fn):
    """Engine entry points run with the engine's GPU current: the C ABI launches on the given stream and reads per-device
    facts (CU count, LDS grants) of the CURRENT device, and a process may hold engines on several GPUs."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        for t in list(args) + list(kwargs.values()):          # device check first: nothing below may touch a host tensor
            if isinstance(t, torch.Tensor):
                HipEngine._dev(self, t, f"a tensor passed to {fn.__name__}")
        with torch.cuda.device(self.device):
            return fn(self, *args, **kwargs)
    return wrapped


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I,H],[I,H] -> [2I,H] with rows 32j..32j+15 = gate[16j..], 32j+16..32j+31 = up[16j..]."""
    I, H = gate.shape
    assert I % 16 == 0
    return torch.stack((gate.view(I // 16, 16, H), up.view(I // 16, 16, H)), dim=1).reshape(2 * I, H).contiguous()


def permute_qk_rows(wqkv: torch.Tensor, n_q: int, n_kv: int, d: int = 128) -> torch.Tensor:
    """Row order coda_qkv_proj_rope expects: inside every q and k head the 128 rows are stored as
    [0..31, 64..95, 32..63, 96..127] (rotary partners j / j+64 land in one MFMA lane); v rows stay."""
    assert d == 128
    order = torch.cat([torch.arange(0, 32), torch.arange(64, 96), torch.arange(32, 64), torch.arange(96, 128)])
    idx = torch.cat([h * d + order for h in range(n_q + n_kv)] + [torch.arange((n_q + n_kv) * d, (n_q + 2 * n_kv) * d)])
    return wqkv[idx.to(wqkv.device)].contiguous()


SPLITK_WS_BYTES = 32 << 20


SPLITK_OVERRIDE: Dict[tuple, int] = {}


def splitk_factor(M: int, N: int, K: int, ws_bytes: int = SPLITK_WS_BYTES) -> int:
    """K ranges for a projection at small M (coda_gemm_bf16_splitk).  A launch with fewer than ~160 tiles of 128 x 128 leaves
    most CUs without work while the busy ones are bound by their own 64 B/clk load path; cutting K engages up to ~450
    workgroups (two per CU).  Returns 1 when the plain engines fill the chip."""
    tiles = -(-M // 128) * -(-N // 128)
    if tiles >= 160 or M > 1024:          # measured at M = 384: gate|up (288 tiles) in two ranges is a loss (4.40 vs 3.81 ms per step)
        return 1
    if (M, N, K) in SPLITK_OVERRIDE:      # tools only (tools/b1_bench.py sweeps): never set by the product path
        return SPLITK_OVERRIDE[(M, N, K)]
    # 257..384 rows (B = 1 at 128 + 256 tokens): when 192 x 96 tiles x ranges fit one workgroup per CU the library writes the
    # partial tiles with its three-stage engine (launch_splitk_partials in gemm_bf16.hip: the same function of M, N, ranges),
    # which needs half the ranges - half the fp32 partial traffic - for the same time in the K loop.  Measured per denoise
    # step (tools/splitk_sweep.sh): o_proj 8 -> 4 ranges 3.27 -> 3.20 ms, qkv 4 -> 2 3.20 -> 3.18; down 8 -> 4 is a loss
    # (3.19: 24 K tiles per workgroup), hence the bound on the range length
    if 256 < M <= 384:
        tall = 2 * -(-N // 96)
        for s in range(256 // tall, 1, -1):
            if tall * s >= 128 and K % s == 0 and (K // s) % 64 == 0 and K // s <= 1024 and s * M * N * 4 <= ws_bytes:
                return s
    best = 1
    for s in range(2, max(1, 448 // tiles) + 1):
        if K % s == 0 and (K // s) % 64 == 0 and K // s >= 256 and s * M * N * 4 <= ws_bytes:
            best = s
    return best


class HipEngine:
    """One engine per model replica (per GPU)."""

    def __init__(self, cfg, state: Dict[str, torch.Tensor], device: torch.device):
        if device.type != "cuda":
            raise _lib.CodaHipError("the CoDA sampler engine runs on an MI355X only (model is on %s)" % device)
        _lib.load()
        self.device = device
        self.V, self.H, self.I = cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size
        self.n_q, self.n_kv, self.d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.layers = cfg.num_hidden_layers
        self.eps = float(cfg.rms_norm_eps)
        self.theta = cfg.rope_theta
        if self.d != 128:
            raise _lib.CodaHipError("HIP kernels are specialised for head_dim 128")
        if self.H % 64 or self.I % 64 or (self.n_q * self.d) % 64:
            raise _lib.CodaHipError("hidden/intermediate sizes must be multiples of 64")
        self.qkv_dim = (self.n_q + 2 * self.n_kv) * self.d
        bf = torch.bfloat16

        def w(key):
            return state[key].detach().to(device=device, dtype=bf).contiguous()

        self.embed = w("model.embed_tokens.weight")
        self.lm_head = w("lm_head.weight")
        self.final_norm = w("model.norm.weight")
        self.lw = []
        for i in range(self.layers):
            p = f"model.layers.{i}."
            self.lw.append(dict(
                n_in=w(p + "input_layernorm.weight"), n_post=w(p + "post_attention_layernorm.weight"),
                qn=w(p + "self_attn.q_norm.weight"), kn=w(p + "self_attn.k_norm.weight"),
                wqkv=torch.cat([w(p + "self_attn.q_proj.weight"), w(p + "self_attn.k_proj.weight"),
                                w(p + "self_attn.v_proj.weight")], dim=0).contiguous(),
                wo=w(p + "self_attn.o_proj.weight"),
                wgu=interleave_gate_up(w(p + "mlp.gate_proj.weight"), w(p + "mlp.up_proj.weight")),
                wd=w(p + "mlp.down_proj.weight")))
        # fused QKV + q/k-norm + RoPE (four-wave engine: a wave's 128 features are one head, weights in their natural row
        # order - no packed copy) needs whole 256-feature tiles inside the q and the k block
        self.fuse_qk = self.n_q % 2 == 0 and self.n_kv % 2 == 0
        self._shape = None
        self._rope = {}
        self.timer = None          # optional KernelTimer (bench.py)
        self.prune_last_layer = True   # forward_hidden(need=...): last layer's per-token tail on the consumed rows only
        self.fuse_splitk_epilogues = True  # small T: split-K reduce kernels also do the row-wise op that follows (A/B knob)
        self.materialise_min_rows = 2048   # launch bound from which lmhead_confidence takes the materialised route
        self.logits_chunk_rows = 8192      # ... and its logits buffer's rows

    # ------------------------------------------------------------------ workspaces
    def _ensure(self, B: int, L: int):
        if self._shape == (B, L):
            return
        T, dev, bf = B * L, self.device, torch.bfloat16
        self.h = torch.empty(T, self.H, device=dev, dtype=bf)
        self.a = torch.empty(T, self.H, device=dev, dtype=bf)
        self.qkv = torch.empty(T, self.qkv_dim, device=dev, dtype=bf)
        self.ao = torch.empty(T, self.n_q * self.d, device=dev, dtype=bf)
        self.mlp = torch.empty(T, self.I, device=dev, dtype=bf)
        self.hn = torch.empty(T, self.H, device=dev, dtype=bf)
        self.src_rows = torch.zeros(T, device=dev, dtype=torch.int32)
        self.dst_pos = torch.zeros(T, device=dev, dtype=torch.int32)
        self.count = torch.zeros(1, device=dev, dtype=torch.int32)
        self.k_out = torch.zeros(1, device=dev, dtype=torch.int32)
        self.conf = torch.zeros(T, device=dev, dtype=torch.float32)
        self.x0 = torch.zeros(T, device=dev, dtype=torch.int64)
        ws = _lib.load().coda_lmhead_workspace_bytes(T, self.V)
        self.lm_ws = torch.empty(ws, device=dev, dtype=torch.uint8)
        if L not in self._rope:
            # cos/sin exactly as the reference builds them: fp32 on the host, then cast to bf16
            inv = 1.0 / (self.theta ** (torch.arange(0, self.d, 2, dtype=torch.int64).float() / self.d))
            ang = torch.arange(L).float()[:, None] * inv[None, :]
            emb = torch.cat((ang, ang), dim=-1)
            self._rope = {L: (emb.cos().to(bf).to(dev).contiguous(), emb.sin().to(bf).to(dev).contiguous())}
        self._shape = (B, L)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _dev(self, t: Optional[torch.Tensor], what: str, dtype=None) -> Optional[torch.Tensor]:
        """Every tensor whose address reaches a kernel must live on this engine's GPU: a host (or other-GPU) pointer
        dereferenced by a kernel is a memory-access fault that takes the process down, where the reference raises a
        Python device-mismatch error.  Checked BEFORE any launch."""
        if t is None:
            return None
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            where = getattr(t, "device", type(t).__name__)
            raise _lib.CodaHipError(f"{what} is on {where}, but the model's HIP engine is on {self.device}: move it with "
                                    f".to('{self.device}') (there is no CPU fallback)")
        if dtype is not None and t.dtype != dtype:
            raise _lib.CodaHipError(f"{what} must be {dtype}, got {t.dtype}")
        return t

    def _gemm(self, name, X, W, C, epi, R=None, rows=None, M=None, split_like=None):
        """``split_like``: row count whose split-K factor this call must use (the compact tail runs on few rows but has to
        sum K in the same order as the full forward would, so that its rows stay bit-identical to it)."""
        M = X.shape[0] if M is None else M
        N, K = W.shape
        ldc = C.stride(0)
        t = self.timer
        if t is not None:
            t.begin(name, 2.0 * M * N * K)
        ks = splitk_factor(M if split_like is None else split_like, N, K) if (rows is None and epi != _lib.EPI_F32 and N % 32 == 0) else 1
        if ks > 1 and ks * M * N * 4 > SPLITK_WS_BYTES:
            ks = 1
        if ks > 1:
            if getattr(self, "_splitk_ws", None) is None:
                self._splitk_ws = torch.empty(SPLITK_WS_BYTES, device=self.device, dtype=torch.uint8)
            _lib.call("coda_gemm_bf16_splitk", _ptr(X), X.stride(0), _ptr(W), W.stride(0), _ptr(C), ldc, _ptr(R),
                      0 if R is None else R.stride(0), M, N, K, epi, ks, _ptr(self._splitk_ws), SPLITK_WS_BYTES, self._stream())
        else:
            _lib.call("coda_gemm_bf16", _ptr(X), X.stride(0), _ptr(rows), _ptr(W), W.stride(0), _ptr(C), ldc,
                      _ptr(R), 0 if R is None else R.stride(0), M, N, K, epi, self._stream())
        if t is not None:
            t.end(name)

    # ------------------------------------------------------------------ decoder forward
    def _tail_rows(self, B: int, L: int, lo: int, hi: int):
        """Token rows b*L + [lo, hi) of every sequence, int32 on the device (rebuilt when the range moves: once per block)."""
        key = (B, L, lo, hi)
        if getattr(self, "_tail_key", None) != key:
            dev, bf, R = self.device, torch.bfloat16, B * (hi - lo)
            self._tail_idx = (torch.arange(B, device=dev, dtype=torch.int32)[:, None] * L
                              + torch.arange(lo, hi, device=dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
            if getattr(self, "_tail_R", None) != R:
                self.h_c = torch.empty(R, self.H, device=dev, dtype=bf)
                self.a_c = torch.empty(R, self.H, device=dev, dtype=bf)
                self.ao_c = torch.empty(R, self.n_q * self.d, device=dev, dtype=bf)
                self.mlp_c = torch.empty(R, self.I, device=dev, dtype=bf)
                self.hn_c = torch.empty(R, self.H, device=dev, dtype=bf)
                self._tail_R = R
            self._tail_key = key
        return self._tail_idx

    @torch.no_grad()
    @_on_device
    def forward_hidden(self, ids: torch.Tensor, need=None) -> torch.Tensor:
        """ids int64 [B,L] on the device -> final-normed hidden states bf16 [B*L, H] (engine-owned).

        ``need=(lo, hi)``: the caller consumes only the rows of positions [lo, hi) of every sequence (the sampler reads
        the hidden state at p-1 for each candidate position p, generation_utils.py:457,462).  The LAST layer's K/V are
        still computed for every token, but its per-token tail (o_proj + residual, post-attention norm, MLP + residual,
        final norm: modeling_coda.py:247-253,353) runs on those rows only, compacted; all other rows of the returned
        buffer are stale.  Row-wise ops: the needed rows come out as in the full forward."""
        self._dev(ids, "input_ids")
        if ids.dtype != torch.int64:
            if ids.dtype.is_floating_point or ids.dtype == torch.bool:
                raise _lib.CodaHipError(f"input_ids must be an integer tensor, got {ids.dtype}")
            ids = ids.to(torch.int64)            # the gather kernel reads 8-byte ids
        B, L = ids.shape
        self._ensure(B, L)
        T, s = B * L, self._stream()
        cos, sin = self._rope[L]
        ids = ids.contiguous()
        # worth it while the two row compactions (2 x 4 KB per kept row) cost less than the skipped rows' GEMMs
        prune = self.prune_last_layer and need is not None and 0 <= need[0] < need[1] <= L and 5 * (need[1] - need[0]) <= 4 * L
        _lib.call("coda_embed_gather", _ptr(ids), _ptr(self.embed), _ptr(self.h), T, self.V, self.H, s)
        scale = 1.0 / math.sqrt(self.d)
        normed = False         # self.a already holds this layer's input norm (written by the previous layer's fused down_proj)
        for li, lw in enumerate(self.lw):
            if not normed:
                _lib.call("coda_rmsnorm", _ptr(self.h), self.H, _ptr(lw["n_in"]), _ptr(self.a), self.H, T, self.H, self.eps, s)
            # the fused epilogue lives on the 256x256 persistent engine: below ~2048 tokens its 256-token tiles leave most
            # of the chip idle (T = 384: 32 workgroups), and the deep-ring 128x128 engine + the standalone norm/rotary kernel win
            if self.fuse_qk and T >= 2048:
                t = self.timer
                if t is not None:
                    t.begin("qkv", 2.0 * T * self.qkv_dim * self.H)
                _lib.call("coda_qkv_proj_rope_w4", _ptr(self.a), self.H, _ptr(lw["wqkv"]), self.H, _ptr(self.qkv), self.qkv_dim,
                          _ptr(lw["qn"]), _ptr(lw["kn"]), _ptr(cos), _ptr(sin), T, L, self.n_q, self.n_kv, self.H, self.eps, s)
                if t is not None:
                    t.end("qkv")
            else:
                ks = self._ksplit(T, self.qkv_dim, self.H) if self.fuse_splitk_epilogues else 1
                if ks > 1:        # small T: split-K partials, reduced by ONE kernel that also does q/k-norm + rotary
                    _lib.call("coda_qkv_proj_rope_splitk", _ptr(self.a), self.H, _ptr(lw["wqkv"]), self.H, _ptr(self.qkv),
                              self.qkv_dim, _ptr(lw["qn"]), _ptr(lw["kn"]), _ptr(cos), _ptr(sin), T, L, self.n_q, self.n_kv, self.H,
                              self.eps, ks, _ptr(self._splitk_workspace()), SPLITK_WS_BYTES, s)
                else:
                    self._gemm("qkv", self.a, lw["wqkv"], self.qkv, _lib.EPI_PLAIN)
                    _lib.call("coda_qknorm_rope", _ptr(self.qkv), self.qkv_dim, _ptr(lw["qn"]), _ptr(lw["kn"]), _ptr(cos),
                              _ptr(sin), T, L, self.n_q, self.n_kv, self.eps, s)
            _lib.call("coda_flash_attn_noncausal", _ptr(self.qkv), self.qkv_dim, _ptr(self.ao), self.n_q * self.d, B, L,
                      self.n_q, self.n_kv, scale, s)
            if prune and li == self.layers - 1:
                return self._compact_tail(lw, B, L, need[0], need[1])
            self._proj_residual_norm("o_proj", self.ao, lw["wo"], self.h, lw["n_post"], self.a, T)
            self._gemm("gate_up", self.a, lw["wgu"], self.mlp, _lib.EPI_SWIGLU)
            last = li == self.layers - 1
            normed = self._proj_residual_norm("down", self.mlp, lw["wd"], self.h, self.final_norm if last else self.lw[li + 1]["n_in"],
                                              self.hn if last else self.a, T, norm_optional=True)
            if last and normed:
                return self.hn
        _lib.call("coda_rmsnorm", _ptr(self.h), self.H, _ptr(self.final_norm), _ptr(self.hn), self.H, T, self.H, self.eps, s)
        return self.hn

    def _ksplit(self, M: int, N: int, K: int) -> int:
        ks = splitk_factor(M, N, K) if N % 32 == 0 else 1
        return ks if ks > 1 and ks * M * N * 4 <= SPLITK_WS_BYTES else 1

    def _splitk_workspace(self) -> torch.Tensor:
        if getattr(self, "_splitk_ws", None) is None:
            self._splitk_ws = torch.empty(SPLITK_WS_BYTES, device=self.device, dtype=torch.uint8)
        return self._splitk_ws

    def _proj_residual_norm(self, name, X, W, h, norm_w, out, M, split_like=None, norm_optional=False) -> bool:
        """h[:M] += X W^T (o_proj / down_proj + residual, modeling_coda.py:247,253), then ``out = RMSNorm(h) * norm_w`` - the
        norm that follows it in the decoder.  With split-K (small M) the partial sums are reduced, added to the residual AND
        normalised by one kernel (coda_gemm_bf16_splitk_norm: same bits, one launch fewer).  ``norm_optional``: when there is
        no split-K the norm is left to the caller (returns False) - the next layer's loop head or the final norm do it."""
        N, K = W.shape
        ks = self._ksplit(M if split_like is None else split_like, N, K)
        if self.fuse_splitk_epilogues and ks > 1 and ks * M * N * 4 <= SPLITK_WS_BYTES and N <= 4096:
            t = self.timer
            if t is not None:
                t.begin(name, 2.0 * M * N * K)
            _lib.call("coda_gemm_bf16_splitk_norm", _ptr(X), X.stride(0), _ptr(W), W.stride(0), _ptr(h), h.stride(0), _ptr(h),
                      h.stride(0), M, N, K, ks, _ptr(self._splitk_workspace()), SPLITK_WS_BYTES, _ptr(norm_w), _ptr(out),
                      out.stride(0), self.eps, self._stream())
            if t is not None:
                t.end(name)
            return True
        self._gemm(name, X, W, h, _lib.EPI_RESIDUAL, R=h, M=M, split_like=split_like)
        if norm_optional:
            return False
        _lib.call("coda_rmsnorm", _ptr(h), h.stride(0), _ptr(norm_w), _ptr(out), out.stride(0), M, self.H, self.eps, self._stream())
        return True

    def _compact_tail(self, lw, B: int, L: int, lo: int, hi: int) -> torch.Tensor:
        """Last layer after attention, on the rows b*L + [lo, hi) only (see forward_hidden)."""
        rows = self._tail_rows(B, L, lo, hi)
        R, s, D = rows.numel(), self._stream(), self.n_q * self.d
        _lib.call("coda_gather_rows", _ptr(self.ao), D, _ptr(rows), None, R, _ptr(self.ao_c), D, D, s)
        _lib.call("coda_gather_rows", _ptr(self.h), self.H, _ptr(rows), None, R, _ptr(self.h_c), self.H, self.H, s)
        T = B * L                                # K is summed in the ranges the full forward uses at T rows: bit-identical rows
        self._proj_residual_norm("o_proj_tail", self.ao_c, lw["wo"], self.h_c, lw["n_post"], self.a_c, R, split_like=T)
        self._gemm("gate_up_tail", self.a_c, lw["wgu"], self.mlp_c, _lib.EPI_SWIGLU, split_like=T)
        self._proj_residual_norm("down_tail", self.mlp_c, lw["wd"], self.h_c, self.final_norm, self.hn_c, R, split_like=T)
        _lib.call("coda_scatter_rows", _ptr(self.hn_c), self.H, _ptr(rows), R, _ptr(self.hn), self.H, B * L, self.H, s)
        return self.hn

    @torch.no_grad()
    @_on_device
    def logits(self, ids: torch.Tensor) -> torch.Tensor:
        """Unshifted logits bf16 [B, L, V] (what model(x) returns in eval mode)."""
        B, L = ids.shape
        hn = self.forward_hidden(ids)
        out = torch.empty(B * L, self.V, device=self.device, dtype=torch.bfloat16)
        self._gemm("lm_head_full", hn, self.lm_head, out, _lib.EPI_PLAIN)
        return out.view(B, L, self.V)

    # ------------------------------------------------------------------ sampler steps
    @_on_device
    def build_candidates(self, x: torch.Tensor, lo: int, hi: int):
        self._dev(x, "x", torch.int64)
        B, L = x.shape
        _lib.call("coda_build_candidates", _ptr(x), B, L, lo, hi, self._mask_id, _ptr(self.src_rows), _ptr(self.dst_pos),
                  _ptr(self.count), self._stream())

    def set_mask_id(self, mask_id: int):
        self._mask_id = int(mask_id)

    # ------------------------------------------------------------------ logits buffers (reserved BEFORE the denoise loop)
    def reserve(self, B: int, L: int, cap: int, filtered: bool, full_logits: bool = False):
        """Called once per generate, before the first denoise step: every buffer the loop will need exists afterwards, so
        no step allocates (a first allocation of 2.5 GB in the middle of a generation can fail after the forward buffers
        are resident).  ``cap``: largest candidate-row bound of the run; ``filtered``: top-p / top-k are active (their rows
        always go through materialised logits); ``full_logits``: a logits hook wants the shifted [B, L, V] tensor."""
        self._ensure(B, L)
        if filtered or cap >= self.materialise_min_rows or full_logits:
            self._reserve_chunk(min(cap, self.logits_chunk_rows))
        if full_logits and getattr(self, "_full_shape", None) != (B, L):
            self._full_hidden = torch.empty(B * L, self.H, device=self.device, dtype=torch.bfloat16)
            self._full_logits = torch.empty(B * L, self.V, device=self.device, dtype=torch.bfloat16)
            pos = torch.arange(L, device=self.device, dtype=torch.int32)
            self._shift_rows = (torch.arange(B, device=self.device, dtype=torch.int32)[:, None] * L
                                + (pos - 1).clamp_min(0)[None, :]).reshape(-1).contiguous()
            self._full_shape = (B, L)

    def _reserve_chunk(self, rows: int):
        """One [rows, V] bf16 logits buffer + its compacted hidden rows, shared by every materialised route (long candidate
        lists, top-p / top-k, hooks).  If the device cannot hold the preferred chunk the chunk is halved (more, shorter
        launches; same results) down to 1024 rows."""
        if getattr(self, "_mat_rows", 0) >= rows:
            return
        self._mat_hidden = self._mat_logits = None
        while True:
            try:
                self._mat_hidden = torch.empty(rows, self.H, device=self.device, dtype=torch.bfloat16)
                self._mat_logits = torch.empty(rows, self.V, device=self.device, dtype=torch.bfloat16)
                break
            except torch.OutOfMemoryError:
                self._mat_hidden = self._mat_logits = None
                if rows <= 1024:
                    raise
                rows = max(1024, rows // 2)
                self.logits_chunk_rows = min(self.logits_chunk_rows, rows)
        self._mat_rows = rows
        if getattr(self, "_mat_counts", None) is None:
            self._mat_counts = torch.zeros(4096, device=self.device, dtype=torch.int32)

    @_on_device
    def lmhead_confidence(self, max_rows: int, alg: int, temperature: float, noise, seed: int, round_conf: bool,
                          top_p=None, top_k=None):
        """lm_head + sample_tokens over the candidate rows (modeling_coda.py:411, generation_utils.py:61-100): the fused
        kernel for short unfiltered lists, the chunked materialised route otherwise."""
        self._dev(noise, "token noise", torch.float32)
        if top_p is not None or top_k is not None or max_rows >= self.materialise_min_rows:
            return self._lmhead_materialised(max_rows, alg, temperature, noise, seed, round_conf, top_p, top_k)
        _lib.call("coda_lmhead_confidence", _ptr(self.hn), self.H, _ptr(self.src_rows), _ptr(self.dst_pos), _ptr(self.count),
                  max_rows, _ptr(self.lm_head), self.H, self.V, self.H, alg, float(temperature), _ptr(noise), seed,
                  int(round_conf), _ptr(self.conf), _ptr(self.x0), _ptr(self.lm_ws), self._stream())

    def _chunks(self, max_rows: int):
        """(chunk rows, number of chunks, per-chunk device counts) for a candidate list walked through the reserved logits
        buffer; the chunks' row counts are derived on the device (no host read)."""
        want = min(max_rows, self.logits_chunk_rows)
        if getattr(self, "_mat_rows", 0) < want:
            self._reserve_chunk(want)                  # direct engine use (tests, tools): reserve() was not called
        chunk = min(want, self._mat_rows)
        n_chunks = (max_rows + chunk - 1) // chunk
        if n_chunks > self._mat_counts.numel():
            raise _lib.CodaHipError(f"{max_rows} candidate rows need {n_chunks} chunks of {chunk} rows: raise logits_chunk_rows")
        if n_chunks > 1:
            _lib.call("coda_chunk_counts", _ptr(self.count), chunk, n_chunks, _ptr(self._mat_counts), self._stream())
        return chunk, n_chunks

    def _lmhead_materialised(self, max_rows: int, alg: int, temperature: float, noise, seed: int, round_conf: bool,
                             top_p=None, top_k=None):
        """Same result as the fused kernel by the route that measures faster from ~2k candidate rows on (17 % at 4k-65k rows,
        `tools/lmhead_unfused_bench.py`), and the only route for top-p / top-k: the rows are compacted, projected by the
        four-wave GEMM engine into a bf16 logits buffer and reduced by the confidence kernel - chunk by chunk, so the
        buffer stays at ``logits_chunk_rows`` x V (2.5 GB) however long the candidate list is."""
        chunk, n_chunks = self._chunks(max_rows)
        s, hid, out = self._stream(), self._mat_hidden, self._mat_logits
        for c in range(n_chunks):
            rows = min(chunk, max_rows - c * chunk)
            cnt = _ptr(self.count) if n_chunks == 1 else _ptr(self._mat_counts) + 4 * c
            _lib.call("coda_gather_rows", _ptr(self.hn), self.hn.stride(0), _ptr(self.src_rows) + 4 * c * chunk, cnt, rows,
                      _ptr(hid), hid.stride(0), self.H, s)
            _lib.call("coda_gemm_bf16_counted", _ptr(hid), hid.stride(0), cnt, _ptr(self.lm_head), self.lm_head.stride(0),
                      _ptr(out), out.stride(0), rows, self.V, self.H, s)
            _lib.call("coda_confidence_from_logits", _ptr(out), out.stride(0), 1, rows, self.V, cnt,
                      _ptr(self.dst_pos) + 4 * c * chunk, alg, float(temperature),
                      float(top_p) if top_p is not None else 0.0, int(top_k) if top_k is not None else 0,
                      None if noise is None else _ptr(noise) + 4 * c * chunk * self.V, seed, int(round_conf),
                      _ptr(self.conf), _ptr(self.x0), s)

    @_on_device
    def candidate_logits(self, max_rows: int) -> torch.Tensor:
        """bf16 logits [max_rows, V] of the candidate rows in the reserved buffer (tests and tools; the denoise loop walks
        the same buffer chunk by chunk in _lmhead_materialised).  max_rows must fit one chunk."""
        self._reserve_chunk(max_rows)
        hc, out = self._mat_hidden, self._mat_logits
        _lib.call("coda_gather_rows", _ptr(self.hn), self.hn.stride(0), _ptr(self.src_rows), _ptr(self.count), max_rows,
                  _ptr(hc), hc.stride(0), self.H, self._stream())
        # only count[0] rows are projected (the candidate set shrinks every step; max_rows bounds the launch)
        _lib.call("coda_gemm_bf16_counted", _ptr(hc), hc.stride(0), _ptr(self.count), _ptr(self.lm_head), self.lm_head.stride(0),
                  _ptr(out), out.stride(0), max_rows, self.V, self.H, self._stream())
        return out[:max_rows]

    @_on_device
    def shifted_logits(self, B: int, L: int) -> torch.Tensor:
        """What a logits hook sees (generation_utils.py:456-460): bf16 [B, L, V] with logits[:, p] = lm_head(hidden[:, p-1])
        (position 0 keeps its own), from the hidden states of the last forward_hidden() over ALL rows.  The shift is a row
        gather of the hidden states in front of the GEMM - no [B, L, V] `torch.cat` copy; the buffer is engine-owned and
        reused every step (reserve(full_logits=True))."""
        if getattr(self, "_full_shape", None) != (B, L):
            self.reserve(B, L, B * L, False, full_logits=True)
        T = B * L
        _lib.call("coda_gather_rows", _ptr(self.hn), self.hn.stride(0), _ptr(self._shift_rows), None, T, _ptr(self._full_hidden),
                  self.H, self.H, self._stream())
        self._gemm("lm_head_full", self._full_hidden, self.lm_head, self._full_logits, _lib.EPI_PLAIN)
        return self._full_logits.view(B, L, self.V)

    @_on_device
    def confidence_from_full_logits(self, logits: torch.Tensor, max_rows: int, alg: int, temperature: float, top_p, top_k,
                                    noise, seed: int, round_conf: bool):
        """sample_tokens over `logits[mask_index]` (generation_utils.py:462) where ``logits`` is the [B, L, V] tensor a
        logits hook returned: the candidate rows are compacted by coda_gather_rows on dst_pos (bf16) chunk by chunk into
        the reserved buffer - no torch index_select copy of the whole candidate set."""
        B, L, V = logits.shape
        flat = logits.reshape(B * L, V)
        if flat.dtype != torch.bfloat16 or flat.stride(1) != 1 or flat.stride(0) % 8 or flat.data_ptr() % 16:
            # a hook that changed the dtype / layout: the confidence kernel takes fp32 or bf16 rows, compacted by torch
            rows = flat.index_select(0, self.dst_pos[:max_rows].long())
            rows = rows if rows.dtype in (torch.bfloat16, torch.float32) else rows.float()
            return self.confidence_from_logits(rows.contiguous(), alg, temperature, top_p, top_k, noise, seed, round_conf)
        chunk, n_chunks = self._chunks(max_rows)
        s, out = self._stream(), self._mat_logits
        for c in range(n_chunks):
            rows = min(chunk, max_rows - c * chunk)
            cnt = _ptr(self.count) if n_chunks == 1 else _ptr(self._mat_counts) + 4 * c
            _lib.call("coda_gather_rows", _ptr(flat), flat.stride(0), _ptr(self.dst_pos) + 4 * c * chunk, cnt, rows, _ptr(out),
                      out.stride(0), V, s)
            _lib.call("coda_confidence_from_logits", _ptr(out), out.stride(0), 1, rows, V, cnt,
                      _ptr(self.dst_pos) + 4 * c * chunk, alg, float(temperature),
                      float(top_p) if top_p is not None else 0.0, int(top_k) if top_k is not None else 0,
                      None if noise is None else _ptr(noise) + 4 * c * chunk * V, seed, int(round_conf),
                      _ptr(self.conf), _ptr(self.x0), s)

    @_on_device
    def confidence_from_logits(self, logits: torch.Tensor, alg: int, temperature: float, top_p, top_k, noise, seed: int,
                               round_conf: bool, use_count: bool = True, scatter: bool = True,
                               conf: Optional[torch.Tensor] = None, x0: Optional[torch.Tensor] = None):
        self._dev(logits, "logits")
        self._dev(noise, "token noise", torch.float32)
        self._dev(conf, "conf", torch.float32)
        self._dev(x0, "x0", torch.int64)
        N, V = logits.shape
        is_bf16 = 1 if logits.dtype == torch.bfloat16 else 0
        if not is_bf16 and logits.dtype != torch.float32:
            raise _lib.CodaHipError("logits must be bf16 or fp32")
        _lib.call("coda_confidence_from_logits", _ptr(logits), logits.stride(0), is_bf16, N, V,
                  _ptr(self.count) if use_count else None, _ptr(self.dst_pos) if scatter else None, alg, float(temperature),
                  float(top_p) if top_p is not None else 0.0, int(top_k) if top_k is not None else 0, _ptr(noise), seed,
                  int(round_conf), _ptr(self.conf if conf is None else conf), _ptr(self.x0 if x0 is None else x0),
                  self._stream())

    @_on_device
    def select_unmask(self, x: torch.Tensor, lo: int, hi: int, ratio: float, take_all: bool, alg_temp: float, noise,
                      seed: int = 0):
        """noise None with alg_temp > 0: the selection race's Exp(1) variates come from the in-kernel counter hash of
        ``seed`` (no torch RNG launch, no [B, L] noise tensor per step)."""
        self._dev(x, "x", torch.int64)
        self._dev(noise, "selection noise", torch.float32)
        B, L = x.shape
        _lib.call("coda_select_unmask", _ptr(x), _ptr(self.conf), _ptr(self.x0), _ptr(self.count), B, L, lo, hi, self._mask_id,
                  float(ratio), int(take_all), float(alg_temp), _ptr(noise), int(seed), _ptr(self.k_out), self._stream())

    @_on_device
    def select_threshold(self, x: torch.Tensor, lo: int, hi: int, k: int, threshold: float):
        """One sequence; returns (revealed, held_back) - the alg's one host read per step (the loop length depends on it)."""
        self._dev(x, "x", torch.int64)
        B, L = x.shape
        if B != 1:
            raise _lib.CodaHipError("confidence_threshold works on one sequence")
        if getattr(self, "_thr_info", None) is None:
            self._thr_info = torch.zeros(2, dtype=torch.int32, device=x.device)
        _lib.call("coda_select_threshold", _ptr(x), _ptr(self.conf), _ptr(self.x0), L, lo, hi, self._mask_id, int(k),
                  float(threshold), _ptr(self._thr_info), self._stream())
        revealed, held = self._thr_info.tolist()
        return revealed, held

    @_on_device
    def origin_transfer(self, x: torch.Tensor, max_rows: int, u: Optional[torch.Tensor], p_transfer: float, seed: int = 0):
        """u None: the uniforms come from the in-kernel counter hash of ``seed``."""
        self._dev(x, "x", torch.int64)
        self._dev(u, "origin uniforms", torch.float32)
        _lib.call("coda_origin_transfer", _ptr(x), _ptr(self.x0), _ptr(self.dst_pos), _ptr(self.count), max_rows, _ptr(u),
                  int(seed), float(p_transfer), self._stream())
