"""ctypes binding of libcoda_hip.so (the C ABI declared in include/coda_hip.h).

There is no fallback: if the library is missing or a call fails, an exception is raised."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcoda_hip.so")

EPI_PLAIN, EPI_RESIDUAL, EPI_SWIGLU, EPI_F32 = 0, 1, 2, 3
GEMM_NARROW_STORES = 1
ALG_IDS = {"maskgit_plus": 0, "topk_margin": 1, "entropy": 2}
ERRORS = {-1: "CODA_ERR_SHAPE", -2: "CODA_ERR_ALIGN", -3: "CODA_ERR_LAUNCH", -4: "CODA_ERR_ARG"}

_p, _i, _ll, _f, _ull = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_ulonglong

# name -> (restype, argtypes); mirrors include/coda_hip.h - tests/test_abi.py parses the header's prototypes and
# compares names, arity and scalar kinds with this table, and checks that the built library exports every symbol
SIGNATURES = {
    "coda_version": (_i, []),
    "coda_last_error": (ctypes.c_char_p, []),
    "coda_temperature_fast_division": (_i, [_f]),
    "coda_embed_gather": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "coda_rmsnorm": (_i, [_p, _ll, _p, _p, _ll, _i, _i, _f, _p]),
    "coda_gemm_bf16": (_i, [_p, _ll, _p, _p, _ll, _p, _ll, _p, _ll, _i, _i, _i, _i, _p]),
    "coda_gemm_bf16_ex": (_i, [_p, _ll, _p, _p, _ll, _p, _ll, _p, _ll, _i, _i, _i, _i, _i, _i, _p]),
    "coda_gemm_bf16_splitk": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _ll, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "coda_gemm_bf16_splitk_norm": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _ll, _i, _i, _i, _i, _p, _ll, _p, _p, _ll, _f, _p]),
    "coda_qkv_proj_rope_splitk": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p, _ll, _p]),
    "coda_qknorm_rope": (_i, [_p, _ll, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "coda_qkv_proj_rope": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "coda_qkv_proj_rope_w4": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "coda_flash_attn_noncausal": (_i, [_p, _ll, _p, _ll, _i, _i, _i, _i, _f, _p]),
    "coda_flash_attn_noncausal_ex": (_i, [_p, _ll, _p, _ll, _i, _i, _i, _i, _f, _i, _i, _p]),
    "coda_build_candidates": (_i, [_p, _i, _i, _i, _i, _ll, _p, _p, _p, _p]),
    "coda_lmhead_workspace_bytes": (_ll, [_i, _i]),
    "coda_lmhead_confidence": (_i, [_p, _ll, _p, _p, _p, _i, _p, _ll, _i, _i, _i, _f, _p, _ull, _i, _p, _p, _p, _p]),
    "coda_lmhead_confidence_noalloc": (_i, [_p, _ll, _p, _p, _p, _i, _p, _ll, _i, _i, _i, _f, _p, _ull, _i, _p, _p, _p]),
    "coda_confidence_from_logits": (_i, [_p, _ll, _i, _i, _i, _p, _p, _i, _f, _f, _i, _p, _ull, _i, _p, _p, _p]),
    "coda_select_unmask": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _ll, _f, _i, _f, _p, _ull, _p, _p]),
    "coda_gather_rows": (_i, [_p, _ll, _p, _p, _i, _p, _ll, _i, _p]),
    "coda_chunk_counts": (_i, [_p, _i, _i, _p, _p]),
    "coda_scatter_rows": (_i, [_p, _ll, _p, _i, _p, _ll, _i, _i, _p]),
    "coda_gemm_bf16_counted":(_i, [_p, _ll, _p, _p, _ll, _p, _ll, _i, _i, _i, _p]),
    "coda_select_threshold": (_i, [_p, _p, _p, _i, _i, _i, _ll, _i, _f, _p, _p]),
    "coda_origin_transfer": (_i, [_p, _p, _p, _p, _i, _p, _ull, _f, _p]),
}

_lib = None


class CodaHipError(RuntimeError):
    pass


def load():
    """Loads the shared library once; raises CodaHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CodaHipError(
            f"{LIB_PATH} not found: build it with `make -C coda_amd/csrc` (or __graft_entry__.build()). "
            "coda_amd has no CPU or PyTorch fallback for the sampler path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def call(name, *args):
    """Calls an int-returning entry point and raises on a negative return code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        detail = lib.coda_last_error().decode()
        if rc != -3 and not detail.startswith(name):      # the string is only fresh for launch errors and worded refusals
            detail = ""
        raise CodaHipError(f"{name} failed: {ERRORS.get(rc, rc)} {detail}")
