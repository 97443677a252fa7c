"""Per-kernel parity of the HIP path (through the C ABI) against the CPU oracle / plain torch fp32.

Integer-valued operands make the GEMM-shaped checks BIT-EXACT (every product and partial sum is
exactly representable), so a wrong MFMA lane map, swizzle or epilogue index cannot hide behind a
tolerance; random-normal operands then check the numerics at realistic scale."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from coda_amd import _lib  # noqa: E402
from oracle import ref_cpu  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def _s():
    return torch.cuda.current_stream().cuda_stream


_KEEP = []


def _p(t):
    """data_ptr of a device tensor; the tensor is kept alive until the test ends (a temporary such as
    x.to(DEV) would otherwise be freed, and its block reused, before the kernel has run)."""
    if t is None:
        return None
    _KEEP.append(t)
    return t.data_ptr()


@pytest.fixture(autouse=True)
def _release_kept_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def gemm(X, W, epi, R=None, rows=None, M=None, out_cols=None, engine=0, flags=0):
    """engine / flags != 0: the tile engine and store width are the CALL's arguments (coda_gemm_bf16_ex); the library
    keeps no selection state."""
    M = X.shape[0] if M is None else M
    N, K = W.shape
    cols = out_cols or (N // 2 if epi == _lib.EPI_SWIGLU else N)
    C = torch.full((M, cols), float("nan"), device=DEV, dtype=torch.float32 if epi == _lib.EPI_F32 else BF)
    if engine or flags:
        _lib.call("coda_gemm_bf16_ex", _p(X), X.stride(0), _p(rows), _p(W), W.stride(0), _p(C), C.stride(0), _p(R),
                  0 if R is None else R.stride(0), M, N, K, epi, engine, flags, _s())
    else:
        _lib.call("coda_gemm_bf16", _p(X), X.stride(0), _p(rows), _p(W), W.stride(0), _p(C), C.stride(0), _p(R),
                  0 if R is None else R.stride(0), M, N, K, epi, _s())
    torch.cuda.synchronize()
    return C


def int_mat(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).to(BF)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 132, 192), (1, 4, 64), (300, 1024, 256)])
def test_gemm_exact_integer(M, N, K):
    X, W = int_mat((M, K), -3, 3, 1), int_mat((N, K), -2, 2, 2)       # asymmetric operands
    ref = X.float() @ W.float().t()
    out = gemm(X.to(DEV), W.to(DEV), _lib.EPI_F32).cpu()
    assert torch.equal(out, ref)
    out = gemm(X.to(DEV), W.to(DEV), _lib.EPI_PLAIN).cpu()
    assert torch.equal(out, ref.to(BF))


@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize("M,N,K,epi", [(512, 512, 64, "f32"), (768, 256, 192, "plain"), (300, 520, 128, "f32"),
                                       (1024, 768, 2048, "res"), (515, 512, 320, "res"), (8192, 5120, 128, "plain"),
                                       (8192 + 40, 4096 + 128, 64, "res")])
def test_gemm_engines_exact_integer(variant, M, N, K, epi, wide):
    """Every tile engine (forced per call through coda_gemm_bf16_ex) on exact integer data, with the interior tiles'
    stores as 16-byte pieces (v_permlane16_swap exchange, the default) and as 8-byte pieces."""
    fl = 0 if wide else _lib.GEMM_NARROW_STORES
    X, W = int_mat((M, K), -3, 3, 21), int_mat((N, K), -2, 2, 22)
    ref = X.float() @ W.float().t()
    assert ref.abs().max() < 2 ** 24
    if epi == "f32":
        assert torch.equal(gemm(X.to(DEV), W.to(DEV), _lib.EPI_F32, engine=variant, flags=fl).cpu(), ref)
    elif epi == "plain":
        assert torch.equal(gemm(X.to(DEV), W.to(DEV), _lib.EPI_PLAIN, engine=variant, flags=fl).cpu(), ref.to(BF))
    else:
        R = int_mat((M, N), -50, 50, 23)
        want = (ref.to(BF).float() + R.float()).to(BF)
        assert torch.equal(gemm(X.to(DEV), W.to(DEV), _lib.EPI_RESIDUAL, R=R.to(DEV), engine=variant, flags=fl).cpu(), want)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
def test_gemm_engines_swiglu(variant):
    from coda_amd.engine import interleave_gate_up
    M, I, K = 520, 384, 256
    Wg, Wu = int_mat((I, K), -1, 1, 31), int_mat((I, K), -2, 2, 32)
    X = int_mat((M, K), -1, 1, 33)
    gate, up = (X.float() @ Wg.float().t()).to(BF), (X.float() @ Wu.float().t()).to(BF)
    ref = torch.nn.functional.silu(gate) * up
    out = gemm(X.to(DEV), interleave_gate_up(Wg, Wu).to(DEV), _lib.EPI_SWIGLU, engine=variant).cpu()
    # integer operands: gate/up are exact, only silu's exp differs between host and device libm
    assert (out.float() - ref.float()).abs().max() <= 2.0 ** -7 * ref.float().abs().max()
    assert (out == ref).float().mean() > 0.99


@pytest.mark.parametrize("M", [384, 300])
def test_gemm_few_rows_picker_takes_the_one_tile_per_cu_engine(M):
    """B = 1 gate|up (a few hundred rows x 12288 features): the picker leaves the 128x128 engine (288 tiles, a second, mostly empty
    round) for the 192x96 tile (256 tiles).  Every engine sums a row's K range in the same order, so on RANDOM bf16 data the
    result must equal the 128x128 engine's bit for bit - which is also how this test sees that the picker moved (timing would
    not be a test): the forced engines 1 and 11 bracket the default call."""
    from coda_amd.engine import interleave_gate_up
    g = torch.Generator().manual_seed(77)
    X = torch.randn(M, 2048, generator=g).to(BF).to(DEV)
    W = interleave_gate_up((torch.randn(6144, 2048, generator=g) * 0.03).to(BF), (torch.randn(6144, 2048, generator=g) * 0.03).to(BF)).to(DEV)
    auto = gemm(X, W, _lib.EPI_SWIGLU)
    assert torch.equal(auto, gemm(X, W, _lib.EPI_SWIGLU, engine=1))
    assert torch.equal(auto, gemm(X, W, _lib.EPI_SWIGLU, engine=11))
    assert torch.equal(auto, gemm(X, W, _lib.EPI_SWIGLU, engine=10))
    R = torch.randn(M, 12288, generator=g).to(BF).to(DEV)
    assert torch.equal(gemm(X, W, _lib.EPI_RESIDUAL, R=R, engine=11), gemm(X, W, _lib.EPI_RESIDUAL, R=R, engine=1))
    assert torch.equal(gemm(X, W, _lib.EPI_F32, engine=10), gemm(X, W, _lib.EPI_F32, engine=1))


@pytest.mark.parametrize("engine", [5, 6, 8, 0])
def test_gemm_residual_in_place_equals_out_of_place(engine):
    """The decoder adds o_proj / down_proj into the residual stream IN PLACE (C aliases R).  The 192x256 engines fetch a
    tile's residual values during the last K tiles of that tile (before any of its stores) - the aliased call must give
    the out-of-place result bit for bit, on random data, full and ragged tiles alike."""
    g = torch.Generator().manual_seed(91)
    M, N, K = 1000, 768, 2048
    X = torch.randn(M, K, generator=g).to(BF).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.03).to(BF).to(DEV)
    R = torch.randn(M, N, generator=g).to(BF).to(DEV)
    want = gemm(X, W, _lib.EPI_RESIDUAL, R=R, engine=1)
    assert torch.equal(gemm(X, W, _lib.EPI_RESIDUAL, R=R, engine=engine), want)
    C = R.clone()
    _lib.call("coda_gemm_bf16_ex", _p(X), K, None, _p(W), K, _p(C), N, _p(C), N, M, N, K, _lib.EPI_RESIDUAL, engine, 0, _s())
    torch.cuda.synchronize()
    assert torch.equal(C, want)


def test_gemm_ex_rejects_unknown_engine():
    X = torch.zeros(64, 64, device=DEV, dtype=BF)
    C = torch.zeros(64, 64, device=DEV, dtype=BF)
    rc = _lib.load().coda_gemm_bf16_ex(_p(X), 64, None, _p(X), 64, _p(C), 64, None, 0, 64, 64, 64, 0, 12, 0, _s())
    assert rc == -4


@pytest.mark.parametrize("M,N,K,ks,epi", [(384, 2048, 2048, 8, "res"), (384, 4096, 2048, 4, "plain"), (384, 2048, 6144, 8, "res"),
                                          (384, 2048, 2048, 4, "res"), (384, 4096, 2048, 2, "plain"), (300, 2048, 6144, 4, "res"),
                                          (200, 256, 512, 2, "plain"), (1024, 2048, 6144, 3, "res"), (130, 384, 1024, 4, "swiglu")])
def test_gemm_splitk_exact_integer(M, N, K, ks, epi):
    """coda_gemm_bf16_splitk (small-M path): fp32 partials of the K ranges summed in a fixed order + the epilogue; integer
    operands make every partial and their sum exact, so the result must equal the one-range engines bit for bit."""
    from coda_amd.engine import interleave_gate_up, splitk_factor
    X = int_mat((M, K), -3, 3, 61).to(DEV)
    ws = torch.empty(ks * M * N * 4, device=DEV, dtype=torch.uint8)
    if epi == "swiglu":
        Wg, Wu = int_mat((N // 2, K), -1, 1, 62), int_mat((N // 2, K), -2, 2, 63)
        W = interleave_gate_up(Wg, Wu).to(DEV)
        want = gemm(X, W, _lib.EPI_SWIGLU, engine=1)
        code, R, cols = _lib.EPI_SWIGLU, None, N // 2
    else:
        W = int_mat((N, K), -2, 2, 62).to(DEV)
        R = int_mat((M, N), -50, 50, 63).to(DEV) if epi == "res" else None
        code, cols = (_lib.EPI_RESIDUAL if epi == "res" else _lib.EPI_PLAIN), N
        want = gemm(X, W, code, R=R, engine=1)
    C = torch.full((M, cols), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_gemm_bf16_splitk", _p(X), K, _p(W), K, _p(C), cols, _p(R), 0 if R is None else N, M, N, K, code, ks, _p(ws),
              ws.numel(), _s())
    torch.cuda.synchronize()
    assert torch.equal(C, want)
    # in place on the residual stream, as the decoder calls it
    if epi == "res":
        Rc = R.clone()
        _lib.call("coda_gemm_bf16_splitk", _p(X), K, _p(W), K, _p(Rc), N, _p(Rc), N, M, N, K, code, ks, _p(ws), ws.numel(), _s())
        torch.cuda.synchronize()
        assert torch.equal(Rc, want)
    assert splitk_factor(384, 2048, 2048) == 4 and splitk_factor(384, 4096, 2048) == 2 and splitk_factor(384, 2048, 6144) == 8
    assert splitk_factor(384, 12288, 2048) == 1 and splitk_factor(12288, 2048, 2048) == 1 and splitk_factor(32, 2048, 2048) == 8
    lib = _lib.load()
    assert lib.coda_gemm_bf16_splitk(_p(X), K, _p(W), K, _p(C), cols, _p(R), 0 if R is None else N, M, N, K, code, ks, _p(ws), 16, _s()) == -1


@pytest.mark.parametrize("M,N,K,ks", [(384, 2048, 2048, 8), (384, 2048, 2048, 4), (384, 2048, 6144, 8), (40, 2048, 1536, 3), (16, 512, 3072, 12), (32, 2048, 6144, 8), (77, 256, 512, 2), (5, 4096, 1024, 4)])
def test_splitk_residual_norm_equals_two_calls(M, N, K, ks):
    """coda_gemm_bf16_splitk_norm (B = 1 serving shape: o_proj / down_proj + residual + the RMSNorm that follows, one reduce
    launch): bit-identical to coda_gemm_bf16_splitk(RESIDUAL) + coda_rmsnorm on random-normal data, out of place and in
    place on the residual stream as the decoder calls it."""
    g = torch.Generator().manual_seed(71)
    X = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
    R = torch.randn(M, N, generator=g).to(BF).to(DEV)
    nw = (1 + 0.2 * torch.randn(N, generator=g)).to(BF).to(DEV)
    ws = torch.empty(ks * M * N * 4 + 64, device=DEV, dtype=torch.uint8)
    C0 = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_gemm_bf16_splitk", _p(X), K, _p(W), K, _p(C0), N, _p(R), N, M, N, K, _lib.EPI_RESIDUAL, ks, _p(ws), ws.numel(), _s())
    A0 = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_rmsnorm", _p(C0), N, _p(nw), _p(A0), N, M, N, 1e-6, _s())
    C1 = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    A1 = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_gemm_bf16_splitk_norm", _p(X), K, _p(W), K, _p(C1), N, _p(R), N, M, N, K, ks, _p(ws), ws.numel(), _p(nw), _p(A1),
              N, 1e-6, _s())
    torch.cuda.synchronize()
    assert torch.equal(C1, C0) and torch.equal(A1, A0)
    Rc, A2 = R.clone(), torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_gemm_bf16_splitk_norm", _p(X), K, _p(W), K, _p(Rc), N, _p(Rc), N, M, N, K, ks, _p(ws), ws.numel(), _p(nw), _p(A2),
              N, 1e-6, _s())
    torch.cuda.synchronize()
    assert torch.equal(Rc, C0) and torch.equal(A2, A0)
    lib = _lib.load()
    assert lib.coda_gemm_bf16_splitk_norm(_p(X), K, _p(W), K, _p(C1), N, _p(R), N, M, N, K, 1, _p(ws), ws.numel(), _p(nw), _p(A1), N, 1e-6, _s()) == -1
    assert lib.coda_gemm_bf16_splitk_norm(_p(X), K, _p(W), K, _p(C1), N, _p(R), N, M, N, K, ks, _p(ws), 16, _p(nw), _p(A1), N, 1e-6, _s()) == -1


@pytest.mark.parametrize("B,L,n_q,n_kv,K,ks", [(1, 384, 16, 8, 2048, 4), (1, 384, 16, 8, 2048, 2), (1, 33, 2, 1, 3072, 12), (2, 50, 4, 2, 512, 2), (1, 33, 2, 1, 256, 2)])
def test_splitk_qkv_rope_equals_two_calls(B, L, n_q, n_kv, K, ks):
    """coda_qkv_proj_rope_splitk: split-K q|k|v projection whose reduce kernel also applies q_norm / k_norm + rotary -
    bit-identical to coda_gemm_bf16_splitk(PLAIN) + coda_qknorm_rope."""
    g = torch.Generator().manual_seed(72)
    T, N = B * L, (n_q + 2 * n_kv) * 128
    X = (torch.randn(T, K, generator=g) * 0.5).to(BF).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
    qw = (1 + 0.2 * torch.randn(128, generator=g)).to(BF).to(DEV)
    kw = (1 + 0.2 * torch.randn(128, generator=g)).to(BF).to(DEV)
    cos, sin = ref_cpu.rope_tables(L, 128, 1e6, BF)
    cos, sin = cos.contiguous().to(DEV), sin.contiguous().to(DEV)
    ws = torch.empty(ks * T * N * 4, device=DEV, dtype=torch.uint8)
    Q0 = torch.full((T, N), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_gemm_bf16_splitk", _p(X), K, _p(W), K, _p(Q0), N, None, 0, T, N, K, _lib.EPI_PLAIN, ks, _p(ws), ws.numel(), _s())
    _lib.call("coda_qknorm_rope", _p(Q0), N, _p(qw), _p(kw), _p(cos), _p(sin), T, L, n_q, n_kv, 1e-6, _s())
    Q1 = torch.full((T, N), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_qkv_proj_rope_splitk", _p(X), K, _p(W), K, _p(Q1), N, _p(qw), _p(kw), _p(cos), _p(sin), T, L, n_q, n_kv, K, 1e-6,
              ks, _p(ws), ws.numel(), _s())
    torch.cuda.synchronize()
    assert torch.equal(Q1, Q0)


def test_gemm_identity_asymmetric():
    K = 128
    X = torch.eye(K).to(BF)
    W = (torch.arange(K * K).reshape(K, K) % 17 - 8).to(BF)             # W[n][k] asymmetric
    out = gemm(X.to(DEV), W.to(DEV), _lib.EPI_F32).cpu()
    assert torch.equal(out, W.float().t())


def test_gemm_residual_and_gather():
    M, N, K = 260, 256, 128
    X, W = int_mat((M, K), -3, 3, 3), int_mat((N, K), -2, 2, 4)
    R = int_mat((M, N), -50, 50, 5)
    ref = ((X.float() @ W.float().t()).to(BF).float() + R.float()).to(BF)
    out = gemm(X.to(DEV), W.to(DEV), _lib.EPI_RESIDUAL, R=R.to(DEV)).cpu()
    assert torch.equal(out, ref)
    # in place (C aliases R), as the decoder uses it
    Rd, Xd, Wd = R.to(DEV).clone(), X.to(DEV), W.to(DEV)          # keep the device operands alive across the launch
    _lib.call("coda_gemm_bf16", _p(Xd), K, None, _p(Wd), K, _p(Rd), N, _p(Rd), N, M, N, K, _lib.EPI_RESIDUAL, _s())
    torch.cuda.synchronize()
    assert torch.equal(Rd.cpu(), ref)
    # row gather
    g = torch.Generator().manual_seed(6)
    rows = torch.randint(0, M, (77,), generator=g).to(torch.int32)
    out = gemm(X.to(DEV), W.to(DEV), _lib.EPI_F32, rows=rows.to(DEV), M=77).cpu()
    assert torch.equal(out, X.float()[rows.long()] @ W.float().t())


@pytest.mark.parametrize("T,H,cap,cnt,N", [(700, 128, 300, 77, 520), (4096, 256, 2048, 1500, 8192), (4096, 256, 2048, 2048, 8192),
                                          (4096, 256, 2048, 1, 8192)])
def test_gather_rows_and_counted_gemm(T, H, cap, cnt, N):
    """`logits[mask_index]` in two steps: compaction of the candidate rows, projection over a device-side row count
    (the second shape takes the persistent 256x256 engine, the first the 128x128 one); integer data: exact."""
    hidden = int_mat((T, H), -3, 3, 51).to(DEV)
    W = int_mat((N, H), -2, 2, 52).to(DEV)
    g = torch.Generator().manual_seed(53)
    rows = torch.randint(0, T, (cap,), generator=g).to(torch.int32).to(DEV)
    count = torch.tensor([cnt], dtype=torch.int32, device=DEV)
    hc = torch.full((cap, H), 7.0, device=DEV, dtype=BF)
    _lib.call("coda_gather_rows", _p(hidden), H, _p(rows), _p(count), cap, _p(hc), H, H, _s())
    out = torch.full((cap, N), 9.0, device=DEV, dtype=BF)
    _lib.call("coda_gemm_bf16_counted", _p(hc), H, _p(count), _p(W), H, _p(out), N, cap, N, H, _s())
    torch.cuda.synchronize()
    assert torch.equal(hc[:cnt].cpu(), hidden.cpu()[rows.cpu().long()[:cnt]])
    assert (hc[cnt:] == 7.0).all()                                   # rows beyond the count are not touched
    ref = (hidden.cpu().float()[rows.cpu().long()[:cnt]] @ W.cpu().float().t()).to(BF)
    assert torch.equal(out[:cnt].cpu(), ref)
    assert (out[cnt:] == 9.0).all()


def test_gemm_swiglu():
    from coda_amd.engine import interleave_gate_up
    M, I, K = 130, 192, 128
    g = torch.Generator().manual_seed(7)
    X = (torch.randn(M, K, generator=g)).to(BF)
    Wg, Wu = (torch.randn(I, K, generator=g) * 0.1).to(BF), (torch.randn(I, K, generator=g) * 0.1).to(BF)
    gate = (X.float() @ Wg.float().t()).to(BF)
    up = (X.float() @ Wu.float().t()).to(BF)
    ref = torch.nn.functional.silu(gate) * up
    out = gemm(X.to(DEV), interleave_gate_up(Wg, Wu).to(DEV), _lib.EPI_SWIGLU).cpu()
    # same rounding points as the reference op sequence; accumulation order may move a value by one bf16 ulp
    err = (out.float() - ref.float()).abs()
    assert (err <= 2.0 ** -7 * ref.float().abs() + 1e-3).all()
    assert (out == ref).float().mean() > 0.98


def test_gemm_random_tolerance():
    M, N, K = 512, 768, 2048
    g = torch.Generator().manual_seed(8)
    X, W = torch.randn(M, K, generator=g).to(BF), (torch.randn(N, K, generator=g) * 0.02).to(BF)
    ref = X.float() @ W.float().t()
    out = gemm(X.to(DEV), W.to(DEV), _lib.EPI_F32).cpu()
    assert (out - ref).abs().max() < 2e-4 * math.sqrt(K)


def test_gemm_rejects_bad_shapes():
    X = torch.zeros(4, 60, device=DEV, dtype=BF)
    W = torch.zeros(8, 60, device=DEV, dtype=BF)
    C = torch.zeros(4, 8, device=DEV, dtype=BF)
    rc = _lib.load().coda_gemm_bf16(_p(X), 60, None, _p(W), 60, _p(C), 8, None, 0, 4, 8, 60, 0, _s())
    assert rc == -1


def test_embed_and_rmsnorm():
    g = torch.Generator().manual_seed(9)
    V, H, T = 300, 256, 37
    table = torch.randn(V, H, generator=g).to(BF)
    ids = torch.randint(0, V, (T,), generator=g)
    out = torch.empty(T, H, device=DEV, dtype=BF)
    _lib.call("coda_embed_gather", _p(ids.to(DEV)), _p(table.to(DEV)), _p(out), T, V, H, _s())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), table[ids])
    for H in (256, 2048, 4608):
        x = (torch.randn(T, H, generator=g) * 3).to(BF)
        w = (1 + 0.1 * torch.randn(H, generator=g)).to(BF)
        y = torch.empty(T, H, device=DEV, dtype=BF)
        _lib.call("coda_rmsnorm", _p(x.to(DEV)), H, _p(w.to(DEV)), _p(y), H, T, H, 1e-6, _s())
        torch.cuda.synchronize()
        ref = ref_cpu._rms(x, w, 1e-6)
        same = (y.cpu() == ref).float().mean().item()
        assert same > 0.995, same                                   # summation order may flip a rare rounding
        assert (y.cpu().float() - ref.float()).abs().max() <= 2.0 ** -6 * ref.float().abs().max()


def test_qknorm_rope():
    g = torch.Generator().manual_seed(10)
    B, L, n_q, n_kv, d = 2, 50, 4, 2, 128
    T = B * L
    qkv = torch.randn(T, (n_q + 2 * n_kv) * d, generator=g).to(BF)
    qw, kw = (1 + 0.1 * torch.randn(d, generator=g)).to(BF), (1 + 0.1 * torch.randn(d, generator=g)).to(BF)
    cos, sin = ref_cpu.rope_tables(L, d, 1000000, BF)
    dq = qkv.to(DEV).clone()
    _lib.call("coda_qknorm_rope", _p(dq), dq.stride(0), _p(qw.to(DEV)), _p(kw.to(DEV)), _p(cos.to(DEV)), _p(sin.to(DEV)),
              T, L, n_q, n_kv, 1e-6, _s())
    torch.cuda.synchronize()
    q = ref_cpu._rms(qkv[:, : n_q * d].view(B, L, n_q, d), qw, 1e-6).transpose(1, 2)
    k = ref_cpu._rms(qkv[:, n_q * d:(n_q + n_kv) * d].view(B, L, n_kv, d), kw, 1e-6).transpose(1, 2)
    q = q * cos[None, None] + ref_cpu._rot_half(q) * sin[None, None]
    k = k * cos[None, None] + ref_cpu._rot_half(k) * sin[None, None]
    ref = qkv.clone()
    ref[:, : n_q * d] = q.transpose(1, 2).reshape(T, n_q * d)
    ref[:, n_q * d:(n_q + n_kv) * d] = k.transpose(1, 2).reshape(T, n_kv * d)
    out = dq.cpu()
    assert torch.equal(out[:, (n_q + n_kv) * d:], qkv[:, (n_q + n_kv) * d:])      # V untouched
    assert (out == ref).float().mean() > 0.995
    assert (out.float() - ref.float()).abs().max() < 0.05


@pytest.mark.parametrize("entry", ["coda_qkv_proj_rope", "coda_qkv_proj_rope_w4"])
@pytest.mark.parametrize("B,L,n_q,n_kv,K", [(2, 200, 4, 2, 256), (3, 128, 2, 2, 128), (1, 300, 16, 8, 2048), (5, 333, 4, 2, 512)])
def test_fused_qkv_rope_equals_two_call_path(B, L, n_q, n_kv, K, entry):
    """coda_qkv_proj_rope (fused epilogue, permuted weight rows) vs coda_gemm_bf16 + coda_qknorm_rope."""
    from coda_amd.engine import permute_qk_rows
    g = torch.Generator().manual_seed(40)
    d, T = 128, B * L
    N = (n_q + 2 * n_kv) * d
    X = torch.randn(T, K, generator=g).to(BF).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(DEV)
    qw, kw = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).to(DEV), (1 + 0.1 * torch.randn(d, generator=g)).to(BF).to(DEV)
    cos, sin = ref_cpu.rope_tables(L, d, 1000000, BF)
    cos, sin = cos.to(DEV), sin.to(DEV)
    ref = gemm(X, W, _lib.EPI_PLAIN)
    _lib.call("coda_qknorm_rope", _p(ref), N, _p(qw), _p(kw), _p(cos), _p(sin), T, L, n_q, n_kv, 1e-6, _s())
    out = torch.full((T, N), float("nan"), device=DEV, dtype=BF)
    if entry == "coda_qkv_proj_rope":
        Wp = permute_qk_rows(W, n_q, n_kv)        # the eight-wave engine wants rotary partners in one MFMA lane
    else:
        Wp = W                                    # four-wave engine: a wave owns a whole head, natural row order
    _lib.call(entry, _p(X), K, _p(Wp), K, _p(out), N, _p(qw), _p(kw), _p(cos), _p(sin), T, L, n_q, n_kv, K, 1e-6, _s())
    torch.cuda.synchronize()
    assert torch.equal(out[:, (n_q + n_kv) * d:], ref[:, (n_q + n_kv) * d:])                 # V: plain projection
    same = (out == ref).float().mean().item()
    err = (out.float() - ref.float()).abs().max().item()
    # identical rounding points; only the order of the 128-term sum of squares differs (rare one-ulp flips)
    assert same > 0.995 and err < 0.05, (same, err)
    # and against the oracle's op sequence on the same GEMM output
    lin = gemm(X, W, _lib.EPI_PLAIN).cpu()
    q = ref_cpu._rms(lin[:, : n_q * d].view(B, L, n_q, d), qw.cpu(), 1e-6).transpose(1, 2)
    q = q * cos.cpu()[None, None] + ref_cpu._rot_half(q) * sin.cpu()[None, None]
    want = q.transpose(1, 2).reshape(T, n_q * d)
    assert (out.cpu()[:, : n_q * d] == want).float().mean() > 0.995


# the last two shapes make a workgroup walk several query blocks (qsplit 1 of 3 blocks with a ragged tail; 2 of 4)
@pytest.mark.parametrize("B,L,n_q,n_kv", [(1, 64, 2, 1), (2, 100, 4, 2), (1, 384, 16, 8), (1, 517, 2, 2), (32, 300, 16, 8),
                                          (16, 512, 16, 8)])
def test_flash_attention(B, L, n_q, n_kv):
    g = torch.Generator().manual_seed(11)
    d, T = 128, B * L
    qkv = torch.randn(T, (n_q + 2 * n_kv) * d, generator=g).to(BF)
    # spike one key against one query so the running max jumps late (online-softmax rescale path)
    qkv[L - 3, :d] *= 6
    qkv[L - 3, n_q * d: n_q * d + d] = qkv[L - 3, :d]
    out = torch.full((T, n_q * d), float("nan"), device=DEV, dtype=BF)
    _lib.call("coda_flash_attn_noncausal", _p(qkv.to(DEV)), qkv.stride(0), _p(out), n_q * d, B, L, n_q, n_kv,
              1.0 / math.sqrt(d), _s())
    torch.cuda.synchronize()
    q = qkv[:, : n_q * d].view(B, L, n_q, d).transpose(1, 2).float()
    k = qkv[:, n_q * d:(n_q + n_kv) * d].view(B, L, n_kv, d).transpose(1, 2).float().repeat_interleave(n_q // n_kv, 1)
    v = qkv[:, (n_q + n_kv) * d:].view(B, L, n_kv, d).transpose(1, 2).float().repeat_interleave(n_q // n_kv, 1)
    ref = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), -1) @ v
    ref = ref.transpose(1, 2).reshape(T, n_q * d)
    err = (out.cpu().float() - ref).abs().max().item()
    assert err < 2e-2, err


@pytest.mark.parametrize("B,L,n_q,n_kv", [(2, 100, 4, 2), (3, 517, 4, 1), (4, 640, 16, 8), (1, 2048, 16, 8)])
def test_flash_attention_head_pairs_bit_identical(B, L, n_q, n_kv):
    """From L = 512 on a workgroup runs the two query heads of a GQA pair on one K/V stream (8 waves); the per-wave
    arithmetic is the same, so both variants must agree bit for bit (pair_mode of coda_flash_attn_noncausal_ex forces
    either), and so must any number of workgroups per (batch, head)."""
    g = torch.Generator().manual_seed(13)
    d, T = 128, B * L
    qkv = torch.randn(T, (n_q + 2 * n_kv) * d, generator=g).to(BF).to(DEV)
    outs = []
    for pair, qsplit in ((0, 0), (1, 0), (-1, 0), (0, 1), (1, 2)):
        out = torch.full((T, n_q * d), float("nan"), device=DEV, dtype=BF)
        _lib.call("coda_flash_attn_noncausal_ex", _p(qkv), qkv.stride(0), _p(out), n_q * d, B, L, n_q, n_kv, 1.0 / math.sqrt(d),
                  pair, qsplit, _s())
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[1].float()).any()
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_flash_attention_exact_uniform():
    """All-equal scores: softmax is uniform, so O = mean(V); with integer V this checks the V transposed-read
    mapping and the P/V k-order pairing without tolerance slack."""
    B, L, n_q, n_kv, d = 1, 128, 2, 1, 128
    T = B * L
    qkv = torch.zeros(T, (n_q + 2 * n_kv) * d)
    g = torch.Generator().manual_seed(12)
    V = torch.randint(-4, 5, (T, d), generator=g).float()
    qkv[:, (n_q + n_kv) * d:] = V
    out = torch.empty(T, n_q * d, device=DEV, dtype=BF)
    _lib.call("coda_flash_attn_noncausal", _p(qkv.to(BF).to(DEV)), qkv.stride(0), _p(out), n_q * d, B, L, n_q, n_kv, 0.1, _s())
    torch.cuda.synchronize()
    ref = V.mean(0, keepdim=True).expand(T, d)
    assert (out.cpu().float()[:, :d] - ref).abs().max() < 2e-2
    # one-hot attention: query i matches key perm[i] only -> O[i] = V[perm[i]] exactly
    perm = torch.randperm(L, generator=g)
    code = torch.zeros(L, d)
    for i in range(L):
        for b in range(7):
            code[i, b] = 1.0 if (i >> b) & 1 else -1.0
    qkv = torch.zeros(T, (n_q + 2 * n_kv) * d)
    qkv[:, :d] = code[perm] * 8
    qkv[:, d:2 * d] = code[perm] * 8
    qkv[:, n_q * d:(n_q + n_kv) * d] = code * 8
    qkv[:, (n_q + n_kv) * d:] = V
    _lib.call("coda_flash_attn_noncausal", _p(qkv.to(BF).to(DEV)), qkv.stride(0), _p(out), n_q * d, B, L, n_q, n_kv, 1.0, _s())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().float()[:, :d], V[perm])
    assert torch.equal(out.cpu().float()[:, d:], V[perm])


def _candidates(x, lo, hi, mask_id):
    B, L = x.shape
    xd = x.to(DEV)
    src = torch.zeros(B * (hi - lo), device=DEV, dtype=torch.int32)
    dst = torch.zeros_like(src)
    cnt = torch.zeros(1, device=DEV, dtype=torch.int32)
    _lib.call("coda_build_candidates", _p(xd), B, L, lo, hi, mask_id, _p(src), _p(dst), _p(cnt), _s())
    torch.cuda.synchronize()
    return src, dst, cnt


def test_build_candidates_order_and_shift():
    g = torch.Generator().manual_seed(13)
    B, L, mask_id = 5, 700, 99
    x = torch.randint(0, 50, (B, L), generator=g)
    x[torch.rand(B, L, generator=g) < 0.4] = mask_id
    x[2, 0] = mask_id
    for lo, hi in ((0, L), (100, 612), (640, 700)):
        src, dst, cnt = _candidates(x, lo, hi, mask_id)
        win = torch.zeros_like(x, dtype=torch.bool)
        win[:, lo:hi] = True
        pos = ((x == mask_id) & win).nonzero()
        n = cnt.item()
        assert n == pos.shape[0]
        flat = pos[:, 0] * L + pos[:, 1]
        assert torch.equal(dst[:n].cpu().long(), flat)
        assert torch.equal(src[:n].cpu().long(), pos[:, 0] * L + (pos[:, 1] - 1).clamp(min=0))


@pytest.mark.parametrize("alg_temp", [0.0, 0.3])
def test_select_unmask_matches_oracle(alg_temp):
    g = torch.Generator().manual_seed(14)
    B, L, P, mask_id = 4, 96, 16, 7
    for trial, (ratio, last) in enumerate(((0.25, False), (0.031, False), (0.5, True), (0.001, False))):
        x = torch.randint(10, 50, (B, L), generator=g)
        x[:, P:][torch.rand(B, L - P, generator=g) < 0.6] = mask_id
        conf = -torch.rand(B, L, generator=g)
        conf[0, P:P + 8] = -0.5                                             # ties: lowest index must win
        x0 = torch.randint(100, 200, (B, L), generator=g)
        noise = torch.empty(B, L).exponential_(1, generator=g)
        lo, hi = P, L
        mask_index = x == mask_id
        n_mask = mask_index.sum() / B
        k = int(n_mask) if last else int(n_mask * torch.tensor(ratio))
        expect = x.clone()
        if k > 0:
            full = torch.full((B, L), -torch.inf)
            full[mask_index] = conf[mask_index]
            key = full if alg_temp == 0 else torch.softmax(full / alg_temp, -1) / noise
            for b in range(B):
                nb = int(mask_index[b].sum())
                order = sorted(range(L), key=lambda p: (-key[b, p].item(), p))[:min(k, nb)]
                order = [p for p in order if mask_index[b, p]]
                expect[b, order] = x0[b, order]
        xd = x.to(DEV)
        src, dst, cnt = _candidates(x, lo, hi, mask_id)
        kd = torch.zeros(1, device=DEV, dtype=torch.int32)
        _lib.call("coda_select_unmask", _p(xd), _p(conf.to(DEV).contiguous()), _p(x0.to(DEV)), _p(cnt), B, L, lo, hi, mask_id,
                  ratio, int(last), alg_temp, _p(noise.to(DEV)), 0, _p(kd), _s())
        torch.cuda.synchronize()
        assert kd.item() == k, (trial, kd.item(), k)
        assert torch.equal(xd.cpu(), expect), trial


from parity import hash_uniform_host  # noqa: E402  (host restatement of coda_hash_uniform)


def test_select_unmask_in_kernel_noise_matches_host_hash():
    """alg_temp > 0 with noise == NULL: the kernel's own Exp(1) variates (-log of the counter hash) must select what
    the reference rule selects with the host restatement of the same hash; other seeds select differently."""
    g = torch.Generator().manual_seed(44)
    B, L, P, mask_id, alg_temp, seed = 6, 200, 20, 7, 0.25, 0x1234_5678_9ABC_DEF1
    x = torch.randint(10, 50, (B, L), generator=g)
    x[:, P:][torch.rand(B, L - P, generator=g) < 0.7] = mask_id
    conf = -torch.rand(B, L, generator=g) * 2
    x0 = torch.randint(100, 200, (B, L), generator=g)
    mask_index = x == mask_id
    pos = torch.arange(B * L).reshape(B, L)
    q = -torch.log(hash_uniform_host(seed, pos, 0xFFFFFFFF))
    assert 0.5 < q.mean() < 1.5 and q.min() > 0                              # Exp(1)-like
    ratio = 0.2
    k = int((mask_index.sum() / B) * torch.tensor(ratio))
    full = torch.full((B, L), -torch.inf)
    full[mask_index] = conf[mask_index]
    key = torch.softmax(full / alg_temp, -1) / q
    expect = x.clone()
    margins = []
    for b in range(B):
        order = sorted(mask_index[b].nonzero().squeeze(1).tolist(), key=lambda p: (-key[b, p].item(), p))
        expect[b, order[:k]] = x0[b, order[:k]]
        margins.append(key[b, order[k - 1]].item() / key[b, order[k]].item() - 1)
    outs = []
    for sd in (seed, seed + 1):
        xd = x.to(DEV)
        src, dst, cnt = _candidates(x, P, L, mask_id)
        _lib.call("coda_select_unmask", _p(xd), _p(conf.to(DEV).contiguous()), _p(x0.to(DEV)), _p(cnt), B, L, P, L, mask_id,
                  ratio, 0, alg_temp, None, sd, None, _s())
        torch.cuda.synchronize()
        outs.append(xd.cpu())
    # device __logf / __expf vs host libm: a decision can only differ where the k-th / (k+1)-th keys are within ~1e-5
    if min(margins) > 1e-4:
        assert torch.equal(outs[0], expect)
    else:
        assert (outs[0] != expect).sum() <= 2
    assert ((outs[0] != x).sum(1) == k).all()
    assert not torch.equal(outs[0], outs[1])


def test_origin_transfer_in_kernel_uniforms():
    g = torch.Generator().manual_seed(45)
    B, L, mask_id, seed = 3, 500, 7, 987654321987
    x = torch.randint(10, 50, (B, L), generator=g)
    x[torch.rand(B, L, generator=g) < 0.5] = mask_id
    x0 = torch.randint(100, 200, (B, L), generator=g)
    src, dst, cnt = _candidates(x, 0, L, mask_id)
    n = cnt.item()
    u = hash_uniform_host(seed, dst[:n].cpu().long(), 0xFFFFFFFE)
    for p_t in (0.3, 1.0, 0.0):
        xd = x.to(DEV)
        _lib.call("coda_origin_transfer", _p(xd), _p(x0.to(DEV)), _p(dst), _p(cnt), B * L, None, seed, p_t, _s())
        torch.cuda.synchronize()
        expect = x.clone().reshape(-1)
        take = dst[:n].cpu().long()[u < p_t]
        expect[take] = x0.reshape(-1)[take]
        assert torch.equal(xd.cpu().reshape(-1), expect), p_t
    assert abs((u < 0.3).float().mean().item() - 0.3) < 0.05


def test_select_threshold_matches_reference_rule():
    """coda_select_threshold vs the serving sampler's rule (serving/dream/model/generation_utils.py:442-465) restated
    with torch.topk on the same confidences: the best of the k is always revealed, the others only at >= threshold."""
    g = torch.Generator().manual_seed(15)
    L, P, mask_id = 160, 24, 7
    for trial, (k, thr) in enumerate(((5, 0.5), (1, 0.99), (12, 0.0), (9, 2.0), (0, 0.5), (40, 0.3))):
        x = torch.randint(10, 50, (1, L), generator=g)
        x[:, P:][torch.rand(1, L - P, generator=g) < 0.5] = mask_id
        conf = torch.rand(1, L, generator=g)
        x0 = torch.randint(100, 200, (1, L), generator=g)
        mask_index = x == mask_id
        k = min(k, int(mask_index.sum()))
        full = torch.full((1, L), -torch.inf)
        full[mask_index] = conf[mask_index]
        sel_conf, sel_idx = torch.topk(full, k)
        transfer = torch.zeros_like(mask_index)
        transfer[0, sel_idx[0]] = True
        held = 0
        for kk in range(1, k):
            if sel_conf[0, kk] < thr:
                held += 1
                transfer[0, sel_idx[0, kk]] = False
        expect = torch.where(transfer, x0, x)
        xd = x.to(DEV)
        info = torch.full((2,), -7, device=DEV, dtype=torch.int32)
        _lib.call("coda_select_threshold", _p(xd), _p(conf.to(DEV).contiguous()), _p(x0.to(DEV)), L, P, L, mask_id, k, thr, _p(info), _s())
        torch.cuda.synchronize()
        assert torch.equal(xd.cpu(), expect), trial
        assert info.tolist() == [int(transfer.sum()), held], (trial, info.tolist(), int(transfer.sum()), held)


def _lmhead(hidden, W, src, dst, cnt, cap, alg, temperature=0.0, noise=None, T_total=None, round_conf=0, seed=0):
    V, H = W.shape
    T_total = T_total or hidden.shape[0]
    conf = torch.full((T_total,), float("nan"), device=DEV)
    x0 = torch.full((T_total,), -1, device=DEV, dtype=torch.int64)
    ws = torch.empty(_lib.load().coda_lmhead_workspace_bytes(cap, V), device=DEV, dtype=torch.uint8)
    _lib.call("coda_lmhead_confidence", _p(hidden), hidden.stride(0), _p(src), _p(dst), _p(cnt), cap, _p(W), W.stride(0), V, H,
              alg, temperature, _p(noise), seed, round_conf, _p(conf), _p(x0), _p(ws), _s())
    torch.cuda.synchronize()
    return conf, x0


@pytest.mark.parametrize("V,n_rows", [(1024, 37), (1536 + 4, 300), (5000, 129)])
def test_lmhead_confidence_exact_logits(V, n_rows):
    """Integer hidden/weights -> the bf16 logits are exact, so x0 must equal the oracle's argmax (first index on
    ties) and the fp32 confidences must agree to the 1e-4 bound of SURVEY.md appendix B."""
    g = torch.Generator().manual_seed(15)
    H, T = 128, 400
    hidden = torch.randint(-2, 3, (T, H), generator=g).to(BF)
    W = torch.randint(-1, 2, (V, H), generator=g).to(BF)
    logits_all = (hidden.float() @ W.float().t()) * 1.0
    assert logits_all.abs().max() < 256
    src = torch.randint(0, T, (n_rows,), generator=g).to(torch.int32)
    dst = torch.randperm(T, generator=g)[:n_rows].to(torch.int32)
    cnt = torch.tensor([n_rows], dtype=torch.int32)
    rows = logits_all[src.long()].to(BF).float() * 0.25            # scale after exact bf16 logits: still exact
    hid_s = (hidden.float() * 0.25).to(BF)                          # 0.25 scaling is exact in bf16
    for alg_id, kw in ((0, {}), (1, dict(margin=True)), (2, dict(neg_entropy=True))):
        c_ref, x_ref = ref_cpu.sample_tokens(rows, 0.0, **kw)
        conf, x0 = _lmhead(hid_s.to(DEV), W.to(DEV), src.to(DEV), dst.to(DEV), cnt.to(DEV), 512, alg_id, T_total=T)
        assert torch.equal(x0.cpu()[dst.long()], x_ref)
        assert (conf.cpu()[dst.long()] - c_ref).abs().max() < 1e-4
        untouched = torch.ones(T, dtype=torch.bool)
        untouched[dst.long()] = False
        assert torch.isnan(conf.cpu()[untouched]).all()


@pytest.mark.parametrize("V,n_rows", [(1536 + 4, 37), (5000, 300), (1024, 5000)])
def test_lmhead_confidence_exact_logits_many_rows(V, n_rows):
    """The same known-answer check with a launch bound of 8192 candidate rows out of a 6000-row hidden buffer (every masked
    position of a long generation goes through the lm-head at once): row counts below one 128-row block and far above it."""
    g = torch.Generator().manual_seed(25)
    H, T = 128, 6000
    hidden = torch.randint(-2, 3, (T, H), generator=g).to(BF)
    W = torch.randint(-1, 2, (V, H), generator=g).to(BF)
    src = torch.randint(0, T, (n_rows,), generator=g).to(torch.int32)
    dst = torch.randperm(T, generator=g)[:n_rows].to(torch.int32)
    cnt = torch.tensor([n_rows], dtype=torch.int32)
    rows = (hidden.float()[src.long()] @ W.float().t()).to(BF).float() * 0.25
    hid_s = (hidden.float() * 0.25).to(BF)
    for alg_id, kw in ((0, {}), (1, dict(margin=True)), (2, dict(neg_entropy=True))):
        c_ref, x_ref = ref_cpu.sample_tokens(rows, 0.0, **kw)
        conf, x0 = _lmhead(hid_s.to(DEV), W.to(DEV), src.to(DEV), dst.to(DEV), cnt.to(DEV), 8192, alg_id, T_total=T)
        assert torch.equal(x0.cpu()[dst.long()], x_ref), alg_id
        assert (conf.cpu()[dst.long()] - c_ref).abs().max() < 1e-4, alg_id
        untouched = torch.ones(T, dtype=torch.bool)
        untouched[dst.long()] = False
        assert torch.isnan(conf.cpu()[untouched]).all()


@pytest.mark.parametrize("temperature", [0.0, 0.7])
def test_lmhead_confidence_independent_of_the_launch_bound(temperature):
    """Full vocabulary, random-normal data, 700 candidate rows: the work split is derived from the row count read on the
    device, not from the launch bound, so any bound >= the count gives the same bits - greedy and with in-kernel noise."""
    g = torch.Generator().manual_seed(26)
    H, V, n_rows = 256, 151936, 700
    hidden = torch.randn(n_rows, H, generator=g).to(BF).to(DEV)
    W = (torch.randn(V, H, generator=g) * 0.25).to(BF).to(DEV)
    src = torch.randperm(n_rows, generator=g).to(torch.int32).to(DEV)
    dst = torch.arange(n_rows, dtype=torch.int32).to(DEV)
    cnt = torch.tensor([n_rows], dtype=torch.int32).to(DEV)
    for alg_id in (0, 1, 2):
        c4, x4 = _lmhead(hidden, W, src, dst, cnt, 4096, alg_id, temperature=temperature, T_total=n_rows, seed=77)
        c1, x1 = _lmhead(hidden, W, src, dst, cnt, 1024, alg_id, temperature=temperature, T_total=n_rows, seed=77)
        assert torch.equal(x4, x1) and torch.equal(c4, c1), alg_id


def test_lmhead_confidence_count_beyond_the_launch_bound_is_clamped():
    """A device-side candidate count LARGER than max_rows (a caller's bookkeeping error) must not write past what max_rows
    sized: the partial records (here: guarded by a canary behind the workspace) and conf / x0 beyond the first max_rows
    candidates stay untouched; the first max_rows candidates come out as in a well-formed call."""
    g = torch.Generator().manual_seed(31)
    H, V, n_rows, cap = 128, 5000, 600, 256
    hidden = torch.randn(n_rows, H, generator=g).to(BF).to(DEV)
    W = (torch.randn(V, H, generator=g) * 0.25).to(BF).to(DEV)
    src = torch.arange(n_rows, dtype=torch.int32).to(DEV)
    dst = torch.arange(n_rows, dtype=torch.int32).to(DEV)
    good = torch.tensor([cap], dtype=torch.int32).to(DEV)
    bad = torch.tensor([n_rows], dtype=torch.int32).to(DEV)
    ws_bytes = _lib.load().coda_lmhead_workspace_bytes(cap, V)
    for alg_id in (0, 2):
        c_ok, x_ok = _lmhead(hidden, W, src, dst, good, cap, alg_id, T_total=n_rows)
        conf = torch.full((n_rows,), float("nan"), device=DEV)
        x0 = torch.full((n_rows,), -1, device=DEV, dtype=torch.int64)
        ws = torch.full((ws_bytes + 4096,), 0x5A, device=DEV, dtype=torch.uint8)
        _lib.call("coda_lmhead_confidence", _p(hidden), hidden.stride(0), _p(src), _p(dst), _p(bad), cap, _p(W), W.stride(0), V, H,
                  alg_id, 0.0, None, 0, 0, _p(conf), _p(x0), _p(ws), _s())
        torch.cuda.synchronize()
        assert torch.equal(conf[:cap], c_ok[:cap]) and torch.equal(x0[:cap], x_ok[:cap])
        assert torch.isnan(conf[cap:]).all() and (x0[cap:] == -1).all()
        assert (ws[ws_bytes:] == 0x5A).all()


def test_in_kernel_noise_top_hash_value_does_not_win():
    """Regression: (seed 77, row key 565, token 148599) hashes to the largest value.  With 24 hash bits + 0.5 that became
    u = 1.0 in fp32, -log(u) = 0 and a race key of +inf: the token won whatever its logit.  u now stays below 1, the key is
    finite, and on a flat row the winner is whoever the host restatement of the hash says."""
    seed, row, V, T = 77, 565, 151936, 0.7
    u = hash_uniform_host(seed, row, torch.arange(V))
    assert int(u.argmax()) == 148599 and float(u.max()) == 1.0 - 2.0 ** -24
    key = -torch.log(-torch.log(u))                      # flat row: the logits add nothing
    assert torch.isfinite(key).all()
    logits = torch.zeros(row + 1, V, dtype=BF)
    logits[row, 148599] = -8.0                           # and this entry in particular cannot win from 8 below the rest
    key[148599] += -8.0 / T
    conf = torch.full((row + 1,), float("nan"), device=DEV)
    x0 = torch.full((row + 1,), -1, device=DEV, dtype=torch.int64)
    ld = logits.to(DEV)
    _lib.call("coda_confidence_from_logits", _p(ld), V, 1, row + 1, V, None, None, 0, T, 0.0, 0, None, seed, 0, _p(conf), _p(x0), _s())
    torch.cuda.synchronize()
    assert int(x0[row]) == int(key.argmax()) != 148599


def test_lmhead_confidence_sampled_with_injected_noise():
    g = torch.Generator().manual_seed(16)
    H, T, V, n_rows = 128, 200, 2048, 70
    hidden = torch.randint(-2, 3, (T, H), generator=g).to(BF)
    W = torch.randint(-1, 2, (V, H), generator=g).to(BF)
    src = torch.arange(n_rows, dtype=torch.int32)
    dst = torch.arange(n_rows, dtype=torch.int32) + 5
    cnt = torch.tensor([n_rows], dtype=torch.int32)
    logits = (hidden.float()[:n_rows] @ W.float().t()).to(BF)
    q = torch.empty(n_rows, V).exponential_(1, generator=g)

    class N:
        def token_exp(self, step, probs):
            return q
    temp = 0.5
    c_ref, x_ref = ref_cpu.sample_tokens(logits.float(), temp, noise=N())
    noise = torch.ones(128, V)
    noise[:n_rows] = q
    for cap in (128, 4096):                  # a block's worth of candidates / a long generation's launch bound
        conf, x0 = _lmhead(hidden.to(DEV), W.to(DEV), src.to(DEV), dst.to(DEV), cnt.to(DEV), cap, 0, temperature=temp,
                           noise=noise.to(DEV), T_total=T)
        assert torch.equal(x0.cpu()[dst.long()], x_ref), cap
        assert (conf.cpu()[dst.long()] - c_ref).abs().max() < 1e-4, cap


def test_lmhead_confidence_full_vocab_random():
    """Real vocabulary size, random-normal weights: compare against sample_tokens on torch-computed bf16 logits.
    x0 may differ only where the oracle's top-2 bf16 logits are within one bf16 ulp (declared near-tie band)."""
    g = torch.Generator().manual_seed(17)
    H, V, n_rows = 256, 151936, 200
    hidden = torch.randn(n_rows, H, generator=g).to(BF)
    W = (torch.randn(V, H, generator=g) * 0.25).to(BF)
    src = torch.arange(n_rows, dtype=torch.int32)
    cnt = torch.tensor([n_rows], dtype=torch.int32)
    logits = (hidden.to(DEV).float() @ W.to(DEV).float().t()).to(BF).float().cpu()
    for alg_id, kw in ((2, dict(neg_entropy=True)), (1, dict(margin=True)), (0, {})):
        c_ref, x_ref = ref_cpu.sample_tokens(logits, 0.0, **kw)
        conf, x0 = _lmhead(hidden.to(DEV), W.to(DEV), src.to(DEV), src.to(DEV), cnt.to(DEV), 256, alg_id, T_total=n_rows)
        top2 = torch.topk(logits, 2, dim=-1)[0]
        ulp = 2.0 ** (torch.floor(torch.log2(top2[:, 0].abs())) - 7)
        near = (top2[:, 0] - top2[:, 1]) <= ulp
        bad = (x0.cpu() != x_ref) & ~near
        assert not bad.any(), (alg_id, int(bad.sum()))
        # confidences are continuous in the logits: a bf16 ulp on a dominant logit moves them by <~ 2^-7 relative
        assert (conf.cpu() - c_ref).abs().max() < 0.05, alg_id
        assert (conf.cpu() - c_ref).abs().median() < 2e-3, alg_id


def _conf_logits(logits, alg, temperature=0.0, top_p=0.0, top_k=0, noise=None, round_conf=0):
    N, V = logits.shape
    conf = torch.full((N,), float("nan"), device=DEV)
    x0 = torch.full((N,), -1, device=DEV, dtype=torch.int64)
    ld = logits.to(DEV).contiguous()
    _lib.call("coda_confidence_from_logits", _p(ld), ld.stride(0), int(ld.dtype == BF), N, V, None, None, alg, temperature,
              top_p, top_k, _p(None if noise is None else noise.to(DEV)), 0, round_conf, _p(conf), _p(x0), _s())
    torch.cuda.synchronize()
    return conf.cpu(), x0.cpu()


@pytest.mark.parametrize("V", [1536, 151936])
def test_confidence_from_logits_all_flavours(V, golden):
    g = torch.Generator().manual_seed(18)
    logits = torch.from_numpy(golden["st_logits"]) if V == 1536 else torch.randn(24, V, generator=g) * 4
    N = logits.shape[0]
    q = torch.empty(N, V).exponential_(1, generator=g)

    class Nz:
        def token_exp(self, step, probs):
            return q
    cases = [dict(), dict(top_p=0.9), dict(top_k=50), dict(top_p=0.7, top_k=30), dict(top_p=0.95, top_k=5),
             dict(temperature=0.7, top_p=0.9), dict(temperature=1.3, top_k=20), dict(temperature=0.5),
             # the lm-eval scripts' setting (eval_mbpp_humaneval.sh: T = 0.1, top_p = 0.9): exact logits in, so this is
             # the tight check of generation_utils.py:61-100 at low temperature (the e2e bands are wide there)
             dict(temperature=0.1), dict(temperature=0.1, top_p=0.9), dict(temperature=0.1, top_k=40),
             dict(temperature=0.1, top_p=0.9, top_k=10)]
    for kw in cases:
        for alg_id, okw in ((0, {}), (1, dict(margin=True)), (2, dict(neg_entropy=True))):
            t = kw.get("temperature", 0.0)
            c_ref, x_ref = ref_cpu.sample_tokens(logits, t, kw.get("top_p"), kw.get("top_k"), noise=Nz(), **okw)
            conf, x0 = _conf_logits(logits, alg_id, t, kw.get("top_p", 0.0), kw.get("top_k", 0), q if t > 0 else None)
            assert torch.equal(x0, x_ref), (kw, alg_id)
            # top-p boundary at V=151936: the reference's fp32 cumsum over ~1e5 sorted probabilities carries
            # ~1e-5 absolute rounding error, as large as the boundary tokens' own probabilities, so the kept sets
            # may differ by a few boundary tokens (each moves the confidence by <~1e-4): declared band 2e-3.
            tol = 2e-3 if (kw.get("top_p") and V > 100000) else 2e-4
            assert (conf - c_ref).abs().max() < tol, (kw, alg_id, (conf - c_ref).abs().max())


def test_confidence_top_p_single_token_nucleus_shortcut():
    """top-p with the row maximum alone above top_p: the kernel knows after its first pass (online softmax) and skips the
    radix select.  Rows on both sides of the boundary, near and far, tied maxima, a banned (-inf) head of the row: tokens and
    confidences as the oracle's sort + cumsum gives them, greedy and sampled, every alg."""
    V, top_p = 4096, 0.9
    rows = []
    for pm in (0.5, 0.85, 0.8995, 0.9005, 0.95, 0.999):
        r = torch.full((V,), -30.0)
        r[1234] = 0.0
        r[77] = math.log((1 - pm) / pm)
        rows.append(r)
    tied = torch.full((V,), -30.0)
    tied[5] = tied[4000] = 0.0                      # two maxima of 0.5 each: never a single-token nucleus
    rows.append(tied)
    banned = torch.full((V,), -30.0)
    banned[:64] = -math.inf
    banned[900] = 3.0
    rows.append(banned)
    logits = torch.stack(rows)
    g = torch.Generator().manual_seed(20)
    q = torch.empty(len(rows), V).exponential_(1, generator=g)

    class Nz:
        def token_exp(self, step, probs):
            return q
    for t in (0.0, 1.0):
        for alg_id, okw in ((0, {}), (1, dict(margin=True)), (2, dict(neg_entropy=True))):
            c_ref, x_ref = ref_cpu.sample_tokens(logits, t, top_p, None, noise=Nz(), **okw)
            conf, x0 = _conf_logits(logits, alg_id, t, top_p, 0, q if t > 0 else None)
            assert torch.equal(x0, x_ref), (t, alg_id, x0.tolist(), x_ref.tolist())
            assert (conf - c_ref).abs().max() < 2e-4, (t, alg_id, conf.tolist(), c_ref.tolist())
    c, _ = _conf_logits(logits, 0, 0.0, top_p, 0, None)
    assert c[3] == 1.0 and c[4] == 1.0 and c[5] == 1.0 and abs(c[2] - 0.8995) < 1e-3 and abs(c[6] - 0.5) < 1e-4 and c[7] == 1.0


def test_confidence_greedy_single_pass_ties_and_banned_entries():
    """Greedy, nothing filtered: the kernel reads the row once (per-thread online softmax, merged).  bf16 rows tie massively:
    the token is the FIRST index of the maximum, the margin is 0 when the maximum is duplicated (anywhere in the row: same
    thread's stride or another's), -inf entries carry nothing; all three confidences against the oracle."""
    g = torch.Generator().manual_seed(23)
    N, V = 12, 151936
    logits = (torch.randn(N, V, generator=g) * 2).to(BF).float()
    logits[0, 777] = logits[0].max()                        # a second maximum somewhere else
    mx1 = logits[1].max()
    logits[1, 5] = mx1
    logits[1, 5 + 256] = mx1                                # ... in the same thread's stride (entries v, v + 256 * k)
    logits[2, :4096] = -math.inf                            # banned head of the row
    logits[3, 100000:] = -math.inf
    logits[4] = 1.5                                         # all equal
    for alg_id, okw in ((0, {}), (1, dict(margin=True)), (2, dict(neg_entropy=True))):
        c_ref, x_ref = ref_cpu.sample_tokens(logits, 0.0, **okw)
        for lg in (logits, logits.to(BF)):
            conf, x0 = _conf_logits(lg, alg_id)
            assert torch.equal(x0, x_ref), alg_id
            assert (conf - c_ref).abs().max() < 2e-4, (alg_id, (conf - c_ref).abs().max())
    c, _ = _conf_logits(logits, 1)
    assert c[0] == 0.0 and c[1] == 0.0 and c[4] == 0.0      # duplicated maximum: margin 0


def test_confidence_from_logits_bf16_ties():
    """bf16 logits tie massively; the kept sets of top-p / top-k must follow the documented rule
    (ties at the top-p boundary enter in index order, ties at the k-th value are all kept)."""
    g = torch.Generator().manual_seed(19)
    N, V = 16, 4096
    logits = (torch.randn(N, V, generator=g) * 2).to(BF)
    lf = logits.float()
    for top_p, top_k in ((0.8, 0), (0.0, 40), (0.6, 25)):
        keep = torch.ones(N, V, dtype=torch.bool)
        if top_p:
            p = torch.softmax(lf, -1)
            order = torch.sort(torch.arange(V)[None].expand(N, V), dim=1)[1]
            # stable descending sort = ties in index order
            srt, idx = torch.sort(lf, dim=-1, descending=True, stable=True)
            ps = p.gather(1, idx)
            excl = torch.cumsum(ps, -1) - ps
            kept_sorted = excl <= top_p
            kept_sorted[:, 0] = True
            keep = torch.zeros(N, V, dtype=torch.bool).scatter_(1, idx, kept_sorted)
        if top_k:
            filt = lf.masked_fill(~keep, torch.finfo(torch.float32).min)
            kth = torch.topk(filt, top_k)[0][:, -1:]
            keep = keep & ~(filt < kth)
        filt = lf.masked_fill(~keep, torch.finfo(torch.float32).min)
        pr = torch.softmax(filt, -1)
        ent_ref = (pr * torch.log(pr + 1e-10)).sum(-1)
        conf, x0 = _conf_logits(logits, 2, 0.0, top_p, top_k)
        assert torch.equal(x0, lf.argmax(-1))
        # the cumulative mass is summed in a different order than torch.cumsum: a boundary tie may flip
        close = (conf - ent_ref).abs() < 2e-4
        assert close.float().mean() >= 0.8, (top_p, top_k, (conf - ent_ref).abs())


@pytest.mark.parametrize("scale,top_p", [(0.05, 0.9), (0.3, 0.95), (2.0, 0.8)])
def test_confidence_top_p_tie_cut_negative_threshold(scale, top_p):
    """Flat bf16 rows (early denoise steps) put the nucleus boundary in the NEGATIVE tail, among many equal logits.
    The reference keeps boundary ties one by one while the running mass <= top_p (top_p_logits, generation_utils.py:39-50;
    stable order = index order here).  maskgit_plus at T = 0 returns 1 / Z_kept, so the kept mass - and from it the number
    of boundary ties kept - can be read back from the confidence and compared with a float64 stable-sort oracle."""
    g = torch.Generator().manual_seed(23)
    N, V = 12, 8192
    logits = (torch.randn(N, V, generator=g) * scale).to(BF)
    lf = logits.double()
    srt, idx = torch.sort(lf, dim=-1, descending=True, stable=True)
    e = torch.exp(srt - srt[:, :1])
    ps = e / e.sum(-1, keepdim=True)
    excl = torch.cumsum(ps, -1) - ps
    kept_sorted = excl <= top_p
    kept_sorted[:, 0] = True
    n_keep = kept_sorted.sum(-1)
    thr = srt.gather(1, (n_keep - 1)[:, None]).squeeze(1)                  # value of the last kept entry
    if scale < 1:
        assert (thr < 0).all(), "the case must put the boundary on a negative logit"
    n_above = (lf > thr[:, None]).sum(-1)
    ties_total = (lf == thr[:, None]).sum(-1)
    ties_kept_ref = n_keep - n_above
    assert (ties_total > ties_kept_ref).any(), "no row cuts inside a tie group: the case does not test the cut"
    z_above = (torch.exp(lf - srt[:, :1]) * (lf > thr[:, None])).sum(-1)
    conf, x0 = _conf_logits(logits, 0, 0.0, top_p, 0)
    z_hip = 1.0 / conf.double()
    ties_kept_hip = (z_hip - z_above) / torch.exp(thr - srt[:, 0])
    print(f"top-p tie cut (scale {scale}): ties at the boundary {ties_total.tolist()}, kept ref {ties_kept_ref.tolist()}, "
          f"kept hip {[round(float(t), 2) for t in ties_kept_hip]}")
    assert torch.equal(x0, lf.argmax(-1))
    # fp32 sums over ~8k kept entries: the tie count read back from 1/conf is good to a fraction of a tie
    assert ((ties_kept_hip - ties_kept_ref.double()).abs() <= 1.25).all()
    # the kept index set: entropy over it must match too (index order of the kept ties does not change the value)
    keep = torch.zeros(N, V, dtype=torch.bool).scatter_(1, idx, kept_sorted)
    pr = torch.softmax(lf.masked_fill(~keep, -1e30), -1)
    ent_ref = (pr * torch.log(pr + 1e-10)).sum(-1)
    conf_e, _ = _conf_logits(logits, 2, 0.0, top_p, 0)
    assert (conf_e.double() - ent_ref).abs().max() < 2e-3


def test_confidence_counting_kernel_equals_generic_kernel():
    """bf16 logits with top-p / top-k go through the counting kernel (one pass over exact 16-bit keys + list work); the
    generic radix kernel stays for fp32 logits, unaligned rows and as the fallback for rows the 16-bit counters cannot hold.
    Same rows through both (the generic one is forced by a 2-byte misaligned copy): tokens identical, confidences to fp32
    summation order - flat, peaked and tie-heavy rows, rows with banned (-inf) entries, and the two fallback triggers (a
    value repeated >= 32 768 times; more distinct values than the list holds) mixed into the same launch."""
    g = torch.Generator().manual_seed(41)
    V = 151936
    rows = [torch.randn(4, V, generator=g) * 0.9, torch.randn(4, V, generator=g) * 7.0,
            (torch.randn(4, V, generator=g) * 2.0 * 4).round() / 4]                       # flat, peaked, ties on a 0.25 grid
    banned = torch.randn(2, V, generator=g) * 3.0
    banned[:, ::3] = -float("inf")
    rep = torch.randn(2, V, generator=g) * 2.0
    rep[:, :40000] = -30.0                                                                # one value 40 000 times: counter wraps
    wide = torch.randn(2, V, generator=g) * torch.exp(torch.randn(2, V, generator=g) * 4.0)   # > 6144 distinct bf16 values
    logits = torch.cat(rows + [banned, rep, wide]).to(BF)
    N = logits.shape[0]
    assert (wide.to(BF)[0].unique().numel() > 6144) and N == 18
    al = logits.to(DEV).contiguous()
    pad = torch.zeros(N, V + 8, dtype=BF, device=DEV)
    pad[:, 1:V + 1] = al
    mis = pad[:, 1:V + 1]                                                                  # same values, rows start 2 bytes off 16
    assert al.data_ptr() % 16 == 0 and mis.data_ptr() % 16 == 2 and mis.stride(0) % 8 == 0

    def run(t, alg, T, top_p, top_k):
        conf = torch.full((N,), float("nan"), device=DEV)
        x0 = torch.full((N,), -7, device=DEV, dtype=torch.int64)
        _lib.call("coda_confidence_from_logits", _p(t), t.stride(0), 1, N, V, None, None, alg, T, top_p, top_k, None, 4242, 0,
                  _p(conf), _p(x0), _s())
        torch.cuda.synchronize()
        return conf.cpu(), x0.cpu()
    for T, top_p, top_k in ((0.0, 0.9, 0), (0.0, 0.0, 50), (0.7, 0.9, 0), (0.1, 0.9, 40), (1.3, 0.0, 20), (0.0, 0.5, 3)):
        for alg in (0, 1, 2):
            ca, xa = run(al, alg, T, top_p, top_k)
            cm, xm = run(mis, alg, T, top_p, top_k)
            assert torch.equal(xa, xm), (T, top_p, top_k, alg, (xa != xm).nonzero().flatten().tolist())
            assert (xa >= 0).all() and (xa < V).all()
            err = (ca - cm).abs()
            assert torch.isfinite(ca).all() and float(err.max()) <= 2e-5 * max(1.0, float(cm.abs().max())), (T, top_p, top_k, alg, err)
    # and against the oracle on the greedy settings (fp32 softmax of the same bf16 values; top-p boundary band as above)
    for top_p, top_k in ((0.9, None), (None, 50)):
        c_ref, x_ref = ref_cpu.sample_tokens(logits[:12].float(), 0.0, top_p, top_k, neg_entropy=True)
        conf, x0 = _conf_logits(logits[:12], 2, 0.0, top_p or 0.0, top_k or 0)
        assert torch.equal(x0, x_ref) and (conf - c_ref).abs().max() < 2e-3, (top_p, top_k)


@pytest.mark.parametrize("scale", [1.0, 12.0])
def test_in_kernel_token_noise_race_matches_host_hash(scale):
    """temperature > 0 with noise == NULL: x0 = argmax(l / T - log(-log u)), u = coda_hash_uniform(seed, row key, v).  The
    kernels skip the hash for entries more than CODA_RACE_WINDOW below the row maximum (they cannot win); the host
    restatement races over EVERY entry: same winners, on flat and on peaked rows, unfused and fused path."""
    g = torch.Generator().manual_seed(71)
    N, V, H, T, seed = 24, 4096, 128, 0.7, 0x0123_4567_89AB_CDEF
    hidden = torch.randint(-2, 3, (N, H), generator=g).to(BF)
    W = (torch.randint(-2, 3, (V, H), generator=g).float() * scale / 8).to(BF)       # exact products: identical logits on both paths
    logits = (hidden.float() @ W.float().t()).to(BF)
    u = hash_uniform_host(seed, torch.arange(N)[:, None], torch.arange(V)[None, :])
    lt = (logits.float() / T).to(BF).float()
    key = lt - torch.log(-torch.log(u))
    want = key.argmax(-1)
    top2 = key.topk(2, -1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3                                        # device __logf vs host log
    conf = torch.full((N,), float("nan"), device=DEV)
    x0 = torch.full((N,), -1, device=DEV, dtype=torch.int64)
    ld = logits.to(DEV)
    _lib.call("coda_confidence_from_logits", _p(ld), V, 1, N, V, None, None, 0, T, 0.0, 0, None, seed, 0, _p(conf), _p(x0), _s())
    torch.cuda.synchronize()
    assert torch.equal(x0.cpu()[clear], want[clear]) and clear.sum() >= N - 2
    p = torch.softmax(lt, -1)
    assert (conf.cpu()[clear] - p[torch.arange(N), want][clear]).abs().max() < 2e-3
    src = torch.arange(N, dtype=torch.int32)
    c2, z2 = _lmhead(hidden.to(DEV), W.to(DEV), src.to(DEV), src.to(DEV), torch.tensor([N], dtype=torch.int32).to(DEV), 128, 0,
                     temperature=T, T_total=N, seed=seed)
    assert torch.equal(z2.cpu()[clear], want[clear])


def _chi2(counts, probs, n):
    exp = probs.double() * n
    keep = exp > 5                                    # pool the far tail into one cell
    c = torch.cat([counts.double()[keep], counts.double()[~keep].sum()[None]])
    e = torch.cat([exp[keep], exp[~keep].sum()[None]])
    if e[-1] == 0:
        c, e = c[:-1], e[:-1]
    return float(((c - e) ** 2 / e).sum()), c.numel() - 1


@pytest.mark.parametrize("top_p,top_k", [(0.0, 0), (0.9, 0), (0.0, 20)])
def test_in_kernel_sampling_follows_the_softmax(top_p, top_k):
    """The production path for temperature > 0 draws its noise in the kernel (counter hash -> exponential race).  The
    other noise tests compare it with a host restatement of the SAME hash, which a statistically poor hash would pass; this
    one asks the question that matters: 131 072 draws (8192 rows x 16 step seeds) from one 512-way distribution must fit
    softmax(l / T) - or its top-p nucleus, renormalised - by chi-square (6 sigma), through the unfused and the fused kernel,
    and the draws of neighbouring rows and of consecutive seeds must be independent (chi-square on coarse pairs)."""
    g = torch.Generator().manual_seed(91)
    V, H, N, T, n_seeds = 512, 64, 8192, 0.7, 16
    hvec = torch.randint(-2, 3, (H,), generator=g).float()
    W = (torch.randint(-2, 3, (V, H), generator=g).float() / 4).to(BF)                 # exact products
    row = (W.float() @ hvec).to(BF)
    lt = (row.float() / T).to(BF).float()
    p = torch.softmax(lt, -1)
    if top_p > 0:
        srt, idx = torch.sort(p, descending=True)
        cum = torch.cumsum(srt, -1)
        keep_sorted = (cum - srt) < top_p                                            # the reference keeps the first token past top_p
        mask = torch.zeros(V, dtype=torch.bool)
        mask[idx[keep_sorted]] = True
        p = torch.where(mask, p, torch.zeros_like(p))
        p = p / p.sum()
    if top_k > 0:                                                                    # ties at the k-th value stay (logits < kth removed)
        kth = torch.topk(lt, top_k).values[-1]
        p = torch.where(lt >= kth, p, torch.zeros_like(p))
        p = p / p.sum()
    logits = row[None, :].expand(N, V).contiguous().to(DEV)
    hidden = hvec.to(BF)[None, :].expand(N, H).contiguous().to(DEV)
    src = torch.arange(N, dtype=torch.int32, device=DEV)
    cnt = torch.tensor([N], dtype=torch.int32, device=DEV)
    draws = {"unfused": [], "fused": []}
    for k in range(n_seeds):
        seed = (0x5EED_0000_0000 + 0x9E3779B97F4A7C15 * (k + 1)) & 0xFFFFFFFFFFFFFFFF     # DeviceNoise.step_seed's stepping
        conf = torch.empty(N, device=DEV)
        x0 = torch.empty(N, device=DEV, dtype=torch.int64)
        _lib.call("coda_confidence_from_logits", _p(logits), V, 1, N, V, None, None, 0, T, top_p, top_k, None, seed, 0, _p(conf), _p(x0), _s())
        torch.cuda.synchronize()
        draws["unfused"].append(x0.cpu())
        if top_p == 0 and top_k == 0:
            _, z = _lmhead(hidden, W.to(DEV), src, src, cnt, N, 0, temperature=T, T_total=N, seed=seed)
            draws["fused"].append(z.cpu())
    for name, d in draws.items():
        if not d:
            continue
        x = torch.stack(d)                                                            # [seeds, rows]
        n = x.numel()
        chi, dof = _chi2(torch.bincount(x.flatten(), minlength=V), p, n)
        assert chi < dof + 6 * (2 * dof) ** 0.5, (name, chi, dof)
        assert (p[x.flatten()] > 0).all(), name                                        # nothing outside the nucleus
        # independence: rank of the token in the sorted distribution, binned into 6 classes of about equal mass
        order = torch.argsort(p, descending=True)
        cls_of = torch.empty(V, dtype=torch.long)
        cls_of[order] = torch.clamp((torch.cumsum(p[order], -1) * 6).long(), max=5)
        c = cls_of[x]
        pc = torch.bincount(cls_of, weights=p.double(), minlength=6).float()
        for a, b, what in ((c[:, :-1], c[:, 1:], "neighbouring rows"), (c[:-1], c[1:], "consecutive seeds")):
            joint = torch.bincount((a * 6 + b).flatten(), minlength=36)
            chi, dof = _chi2(joint, (pc[:, None] * pc[None, :]).flatten(), a.numel())
            assert chi < dof + 6 * (2 * dof) ** 0.5, (name, what, chi, dof)


def test_in_kernel_selection_and_origin_noise_distributions():
    """Same question for the other two in-kernel noise streams.  alg_temp > 0: one position is revealed per row (k = 1), and
    over 16 384 rows x 8 seeds the revealed position must follow softmax(confidence / alg_temp) (generation_utils.py:523-527:
    multinomial without replacement == this race).  origin: each candidate is taken with probability p_transfer."""
    B, L, mask_id, alg_temp = 16384, 8, 7, 0.5
    conf_row = torch.tensor([-0.1, -0.9, -0.3, -2.0, -0.5, -1.4, -0.2, -0.7])
    p = torch.softmax(conf_row / alg_temp, -1)
    x = torch.full((B, L), mask_id, dtype=torch.int64)
    x0 = (100 + torch.arange(L))[None, :].expand(B, L).contiguous()
    conf = conf_row[None, :].expand(B, L).contiguous().to(DEV)
    src, dst, cnt = _candidates(x, 0, L, mask_id)
    picks = []
    for k in range(8):
        xd = x.to(DEV)
        _lib.call("coda_select_unmask", _p(xd), _p(conf), _p(x0.to(DEV)), _p(cnt), B, L, 0, L, mask_id, 0.13, 0, alg_temp, None,
                  (0xABCD_0000 + 0x9E3779B97F4A7C15 * (k + 1)) & 0xFFFFFFFFFFFFFFFF, None, _s())
        torch.cuda.synchronize()
        changed = xd.cpu() != mask_id
        assert (changed.sum(1) == 1).all()
        picks.append(changed.float().argmax(1))
    picks = torch.stack(picks)
    chi, dof = _chi2(torch.bincount(picks.flatten(), minlength=L), p, picks.numel())
    assert chi < dof + 6 * (2 * dof) ** 0.5, (chi, dof)
    for a, b, what in ((picks[:, :-1], picks[:, 1:], "neighbouring rows"), (picks[:-1], picks[1:], "consecutive seeds")):
        chi, dof = _chi2(torch.bincount((a * L + b).flatten(), minlength=L * L), (p[:, None] * p[None, :]).flatten(), a.numel())
        assert chi < dof + 6 * (2 * dof) ** 0.5, (what, chi, dof)
    n = B * L
    for p_t in (0.1, 0.5):
        xd = x.to(DEV)
        _lib.call("coda_origin_transfer", _p(xd), _p(x0.to(DEV)), _p(dst), _p(cnt), n, None, 0x77AA_55 + int(p_t * 100), p_t, _s())
        torch.cuda.synchronize()
        taken = (xd.cpu() != mask_id)
        assert abs(taken.float().mean().item() - p_t) < 6 * (p_t * (1 - p_t) / n) ** 0.5, p_t
        per_pos = taken.float().mean(0)                                                # no position favoured
        assert (per_pos - p_t).abs().max() < 6 * (p_t * (1 - p_t) / B) ** 0.5, p_t
