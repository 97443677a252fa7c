"""coda_lmhead_confidence_noalloc (the entry point that allocates nothing: static device scratch instead of a caller
workspace) against coda_lmhead_confidence, through the C ABI: bit-identical confidences and tokens on two calls in a
row, the second with a smaller input; sizes beyond the static scratch are refused with a worded error."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from coda_amd import _lib  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def _both(hidden, W, src, dst, cnt, cap, alg, temperature, seed, T_total):
    V, H = W.shape
    s = torch.cuda.current_stream().cuda_stream
    lib = _lib.load()
    outs = []
    ws = torch.empty(lib.coda_lmhead_workspace_bytes(cap, V), device=DEV, dtype=torch.uint8)
    for noalloc in (False, True):
        conf = torch.full((T_total,), float("nan"), device=DEV)
        x0 = torch.full((T_total,), -1, device=DEV, dtype=torch.int64)
        args = [hidden.data_ptr(), hidden.stride(0), src.data_ptr(), dst.data_ptr(), cnt.data_ptr(), cap, W.data_ptr(), W.stride(0),
                V, H, alg, temperature, None, seed, 0, conf.data_ptr(), x0.data_ptr()]
        if noalloc:
            before = torch.cuda.memory_allocated()
            _lib.call("coda_lmhead_confidence_noalloc", *args, s)
            assert torch.cuda.memory_allocated() == before
        else:
            _lib.call("coda_lmhead_confidence", *args, ws.data_ptr(), s)
        torch.cuda.synchronize()
        outs.append((conf.cpu(), x0.cpu()))
    return outs


def _case(seed, T, H, V, n_rows, cap):
    g = torch.Generator().manual_seed(seed)
    hidden = torch.randn(T, H, generator=g).to(BF).to(DEV)
    W = (torch.randn(V, H, generator=g) * 0.3).to(BF).to(DEV)
    src = torch.randint(0, T, (cap,), generator=g).to(torch.int32).to(DEV)
    dst = torch.randperm(T, generator=g)[:cap].to(torch.int32).to(DEV)
    cnt = torch.tensor([n_rows], dtype=torch.int32, device=DEV)
    return hidden, W, src, dst, cnt


def test_noalloc_equals_workspace_entry_point_twice_second_smaller():
    # first call: many rows and the whole vocabulary split over the grid; second, smaller call right after it on the same
    # GPU's scratch (stale records of the first call lie next to the second call's): both must equal the workspace path
    for (T, H, V, n_rows, cap), algs in (((3000, 256, 151936, 2500, 2560), ((2, 0.0), (0, 0.7))),
                                         ((700, 256, 5000 + 4, 77, 128), ((2, 0.0), (1, 0.0), (0, 0.7)))):
        hidden, W, src, dst, cnt = _case(61, T, H, V, n_rows, cap)
        for alg, temp in algs:
            (c_ws, x_ws), (c_na, x_na) = _both(hidden, W, src, dst, cnt, cap, alg, temp, 0xABCDEF12345, T)
            written = torch.zeros(T, dtype=torch.bool)
            written[dst[:n_rows].cpu().long()] = True
            assert torch.equal(x_ws, x_na), (V, alg, temp)
            assert torch.equal(c_ws[written], c_na[written]), (V, alg, temp)          # bit-exact: same kernels, same order
            assert torch.isnan(c_na[~written]).all() and (x_na[~written] == -1).all()
            assert not torch.isnan(c_na[written]).any() and (x_na[written] >= 0).all() and (x_na[written] < V).all()


def test_noalloc_refuses_sizes_beyond_its_scratch():
    lib = _lib.load()
    hidden, W, src, dst, cnt = _case(62, 64, 128, 1024, 10, 64)
    conf = torch.zeros(64, device=DEV)
    x0 = torch.zeros(64, device=DEV, dtype=torch.int64)
    with pytest.raises(_lib.CodaHipError, match="exceeds the static scratch"):
        _lib.call("coda_lmhead_confidence_noalloc", hidden.data_ptr(), 128, src.data_ptr(), dst.data_ptr(), cnt.data_ptr(),
                  131072 + 1, W.data_ptr(), 128, 1024, 128, 0, 0.0, None, 0, 0, conf.data_ptr(), x0.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert lib.coda_lmhead_workspace_bytes(131072, 151936) <= 131072 * 32 + 256     # what the static array holds
