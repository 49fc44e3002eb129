"""ctypes binding of librevision_hip.so (the C ABI declared in include/revision_hip.h).

There is NO CPU fallback: if the library is missing or fails to load, every compute entry point
raises.  torch is used only as the owner of device memory and streams; the library sees raw device
pointers and a ``hipStream_t``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# REVISION_HIP_LIB: load another build of the same ABI (A/B kernel measurements); default is the in-tree library.
LIB_PATH = os.environ.get("REVISION_HIP_LIB") or os.path.join(_HERE, "librevision_hip.so")

RV_F32, RV_BF16, RV_I32, RV_I64, RV_U8 = 0, 1, 2, 3, 4
RV_ACT_NONE, RV_ACT_RELU, RV_ACT_SILU_MUL, RV_ACT_QUICK_GELU = 0, 1, 2, 3
RV_FEAT_CLS, RV_FEAT_ALL = 0, 2
TOPK_CAP = 64

_DT = {torch.float32: RV_F32, torch.bfloat16: RV_BF16, torch.int32: RV_I32, torch.int64: RV_I64, torch.uint8: RV_U8}


class RvConfig(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("inter", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("vocab", C.c_int32), ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("adapter_dim", C.c_int32),
                ("adapter_heads", C.c_int32), ("adapter_ff", C.c_int32), ("adapter_layers", C.c_int32),
                ("adapter_text", C.c_int32)]


class HipLibraryError(RuntimeError):
    pass


_lib = None

_p, _i32, _i64, _f, _sz, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_uint64

#: every symbol include/revision_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rv_abi_version": (C.c_int, []),
    "rv_last_error": (C.c_int, [C.c_char_p, _sz]),
    "rv_ctx_create": (C.c_int, [C.POINTER(RvConfig), C.POINTER(_p)]),
    "rv_ctx_destroy": (None, [_p]),
    "rv_weights_bind": (C.c_int, [_p, C.c_char_p, _p, C.c_int, _i64]),
    "rv_init_hash": (C.c_int, [_p, C.c_int, _i64, _u64, _f, _f, _p]),
    "rv_ctx_set_option": (C.c_int, [_p, C.c_char_p, _i64]),
    "rv_ctx_get_option": (C.c_int, [_p, C.c_char_p, C.POINTER(_i64)]),
    "rv_gemm_ws_bytes": (_sz, []),
    "rv_gemm_rows_ws_bytes": (_sz, []),
    "rv_gemm_rows": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, C.c_int, C.c_int, _p]),
    "rv_gemm": (C.c_int, [_p, _p, _i64, _p, _i64, C.c_int, _p, _p, _i64, _p, _i64, C.c_int, C.c_int, _i64, _i64, _i64, _p, _sz, _p]),
    "rv_rmsnorm_quant_fp8": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _f, _p]),
    "rv_quant_rows_fp8": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _p]),
    "rv_gemm_fp8": (C.c_int, [_p, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, C.c_int, C.c_int, _i64, _i64, _i64, _p, C.c_size_t, _p]),
    "rv_gemv_fp8": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _p, _i64, C.c_int, C.c_int, _i64, _i64, _i64, _p]),
    "rv_layernorm": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p]),
    "rv_rmsnorm": (C.c_int, [_p, _p, _p, _i64, _i32, _f, _p]),
    "rv_sine_pos": (C.c_int, [_p, _i32, _i32, _p]),
    "rv_attention": (C.c_int, [_p, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i32, _i32,
                               _i32, _i32, _i32, _i32, _i32, _i32, _f, _p]),
    "rv_project_dense": (C.c_int, [_p, _p, _p, C.c_int, _i64, _p]),
    "rv_clip_encoder_ws_bytes": (_sz, [_p, _i32, _i32, _i32, _i32]),
    "rv_clip_encoder": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "rv_splice_embed": (C.c_int, [_p, _p, _p, _p, _i64, _p]),
    "rv_kv_bytes": (_sz, [_p, _i32, _i32]),
    "rv_llm_ws_bytes": (_sz, [_p, _i32, _i32]),
    "rv_llm_forward": (C.c_int, [_p, _p, _i32, _i32, _i32, _p, _i32, _p, _p, _sz, _p]),
    "rv_llm_layers": (C.c_int, [_p, _p, _i32, _i32, _i32, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "rv_llm_prefill_shared_ws_bytes": (_sz, [_p, _i32, _i32, _i32]),
    "rv_llm_prefill_shared": (C.c_int, [_p, _p, _i32, _i32, _i32, _p, _i32, _p, _p, _sz, _p]),
    "rv_llm_prefill_pool": (C.c_int, [_p, _p, _i32, _i32, _i32, _p, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "rv_llm_prefill_pool_groups": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p, _i32, _p, _i32, _p, _p, _sz, _p]),
    "rv_llm_prefill_pool_groups_ragged": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p, _i32, _p, _i32, _p, _p, _p, _sz, _p]),
    "rv_llm_decode_rows": (C.c_int, [_p, _p, _i32, _p, _p, _i32, _p, _p, _sz, _p]),
    "rv_llm_decode_rows_shared": (C.c_int, [_p, _p, _i32, _p, _p, _p, _i32, _p, _p, _sz, _p]),
    "rv_sample": (C.c_int, [_p, _p, _i32, _i32, _p, _i32, _f, _i32, _f, _p, _p, _p, _p, _p, _p, _p]),
    "rv_entropy_stats": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "rv_topk_cosine": (C.c_int, [_p, C.c_int, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "rv_topk_pool": (C.c_int, [_p, C.c_int, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
}


def lib():
    """Load (once) and return the ctypes handle; raises HipLibraryError when the extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -m revisionllm_amd.build` (hipcc, gfx950). "
                "revisionllm_amd has no CPU fallback.")
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:
            raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().rv_last_error(buf, 512)
    return buf.value.decode()


def check(rc, what):
    if rc != 0:
        raise HipLibraryError(f"{what} failed (status {rc}): {last_error()}")


OPTION_KEYS = ("gemm_tile_variant", "gemm_cus", "gemm_arows", "fp8_decode", "fp8_prefill", "sample_variant", "rows_fill", "rows_spread", "rows_persistent", "rows_single", "gemm_waves", "gemm_mhalf", "precision", "lm_head_split")


class Options:
    """Owner of an OPTIONS-ONLY ``rv_ctx`` (``rv_ctx_create(NULL, ...)``): tunables for the building-block entry points that take
    an optional context (``rv_gemm``, ``rv_gemm_fp8``, ``rv_sample``).  ``Options(gemm_tile_variant=6)``; pass it as ``ctx=``."""

    def __init__(self, **kw):
        self._ctx = C.c_void_p()
        check(lib().rv_ctx_create(None, C.byref(self._ctx)), "rv_ctx_create(options)")
        for k, v in kw.items():
            self.set(k, v)

    def set(self, key, value):
        check(lib().rv_ctx_set_option(self._ctx, key.encode(), int(value)), f"rv_ctx_set_option({key})")
        return self

    def get(self, key):
        v = _i64()
        check(lib().rv_ctx_get_option(self._ctx, key.encode(), C.byref(v)), f"rv_ctx_get_option({key})")
        return int(v.value)

    def __del__(self):
        try:
            if self._ctx and self._ctx.value:
                lib().rv_ctx_destroy(self._ctx)
                self._ctx = C.c_void_p()
        except Exception:
            pass


def ctx_ptr(ctx):
    """``rv_ctx*`` of an ``Options`` / ``Engine`` (or None -> NULL: the library's default tunables)."""
    return None if ctx is None else ctx._ctx


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must live on the GPU and be contiguous."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("librevision_hip needs device tensors (got a CPU tensor); there is no CPU path")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    return _DT[t.dtype]
