"""ctypes binding of librevision_hip.so (the C ABI declared in include/revision_hip.h).

There is NO CPU fallback: if the library is missing or fails to load, every compute entry point
raises.  torch is used only as the owner of device memory and streams; the library sees raw device
pointers and a ``hipStream_t``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# The library exists in two operand FLAVOURS with one ABI (include/revision_hip.h "dtypes"): fp16 operands (the default: the checkpoints'
# own storage type, 11 significand bits) and bf16 operands (the reference's GPU dtype).  An Engine is built for one flavour and calls that
# library; the functional wrappers in ops.py pick the library from the dtype of the 16-bit tensors they are handed.
# REVISION_HIP_LIB / REVISION_HIP_LIB_BF16: load another build of the same ABI (A/B kernel measurements); default: the in-tree libraries.
LIB_PATHS = {"f16": os.environ.get("REVISION_HIP_LIB") or os.path.join(_HERE, "librevision_hip.so"),
             "bf16": os.environ.get("REVISION_HIP_LIB_BF16") or os.path.join(_HERE, "librevision_hip_bf16.so")}
LIB_PATH = LIB_PATHS["f16"]
OP_DTYPES = {"f16": torch.float16, "bf16": torch.bfloat16}

RV_F32, RV_BF16, RV_I32, RV_I64, RV_U8, RV_F16 = 0, 1, 2, 3, 4, 5
RV_ACT_NONE, RV_ACT_RELU, RV_ACT_SILU_MUL, RV_ACT_QUICK_GELU = 0, 1, 2, 3
RV_FEAT_CLS, RV_FEAT_ALL = 0, 2
TOPK_CAP = 64

_DT = {torch.float32: RV_F32, torch.bfloat16: RV_BF16, torch.float16: RV_F16, torch.int32: RV_I32, torch.int64: RV_I64, torch.uint8: RV_U8}


class RvConfig(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("inter", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("vocab", C.c_int32), ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("adapter_dim", C.c_int32),
                ("adapter_heads", C.c_int32), ("adapter_ff", C.c_int32), ("adapter_layers", C.c_int32),
                ("adapter_text", C.c_int32)]


class HipLibraryError(RuntimeError):
    pass


_libs = {}
_flavour = os.environ.get("REVISION_OP_DTYPE", "f16")
if _flavour not in OP_DTYPES:
    raise ValueError(f"REVISION_OP_DTYPE={_flavour!r}: expected one of {sorted(OP_DTYPES)}")


def flavour():
    """The process-wide DEFAULT flavour ("f16" unless REVISION_OP_DTYPE / set_flavour say otherwise): what an Engine built without an
    explicit ``op_dtype`` and the wrappers without a 16-bit input tensor use."""
    return _flavour


def set_flavour(f):
    global _flavour
    f = flavour_of(f)
    prev, _flavour = _flavour, f
    return prev


def flavour_of(x):
    """"f16" / "bf16" from a flavour name, a torch dtype or a tensor (None -> the default flavour)."""
    if x is None:
        return _flavour
    if isinstance(x, str):
        if x not in OP_DTYPES:
            raise ValueError(f"unknown operand flavour {x!r}")
        return x
    dt = x.dtype if torch.is_tensor(x) else x
    for k, v in OP_DTYPES.items():
        if v == dt:
            return k
    raise HipLibraryError(f"{dt} is not an operand type of librevision_hip (fp16 or bf16)")


def op_dtype(f=None):
    """torch dtype of a flavour's 16-bit operands."""
    return OP_DTYPES[flavour_of(f)]

_p, _i32, _i64, _f, _sz, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_uint64

#: every symbol include/revision_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rv_abi_version": (C.c_int, []),
    "rv_operand_dtype": (C.c_int, []),
    "rv_last_error": (C.c_int, [C.c_char_p, _sz]),
    "rv_ctx_create": (C.c_int, [C.POINTER(RvConfig), C.POINTER(_p)]),
    "rv_ctx_destroy": (None, [_p]),
    "rv_weights_bind": (C.c_int, [_p, C.c_char_p, _p, C.c_int, _i64]),
    "rv_init_hash": (C.c_int, [_p, C.c_int, _i64, _u64, _f, _f, _p]),
    "rv_ctx_set_option": (C.c_int, [_p, C.c_char_p, _i64]),
    "rv_ctx_get_option": (C.c_int, [_p, C.c_char_p, C.POINTER(_i64)]),
    "rv_numeric_status_bind": (C.c_int, [_p]),
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
    "rv_sample": (C.c_int, [_p, _p, _i32, _i32, _p, _i32, _f, _i32, _f, _p, _p, _p, _p, _p, _p, _p, _p]),
    "rv_entropy_stats": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "rv_topk_cosine": (C.c_int, [_p, C.c_int, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "rv_topk_pool": (C.c_int, [_p, C.c_int, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
}


def lib(f=None):
    """Load (once) and return the ctypes handle of a flavour's library (f: flavour name, torch dtype, tensor or None = the default
    flavour); raises HipLibraryError when the extension is missing."""
    f = flavour_of(f)
    h = _libs.get(f)
    if h is None:
        path = LIB_PATHS[f]
        if not os.path.exists(path):
            raise HipLibraryError(
                f"{path} not found: build it with `python -m revisionllm_amd.build` (hipcc, gfx950). "
                "revisionllm_amd has no CPU fallback.")
        try:
            h = C.CDLL(path)
        except OSError as e:
            raise HipLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.rv_operand_dtype() != _DT[OP_DTYPES[f]]:
            raise HipLibraryError(f"{path} is not the {f} build of the library (rv_operand_dtype() = {h.rv_operand_dtype()})")
        _libs[f] = h
    return h


_numeric = {}


def numeric_status(f=None, device=None):
    """The numeric status buffer of a flavour's library on a device (``rv_numeric_status_bind``): int32 [4] device tensor, word 0 = the sticky count of
    f32 -> fp16 stores that saturated.  Created, zeroed and bound on first use (one per library instance and device; the library keeps the pointer, this
    module keeps the tensor alive).  Reading it is an ordinary device -> host copy: pipelines that copy results out anyway take it along."""
    f = flavour_of(f)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (f, dev.index)
    t = _numeric.get(key)
    if t is None:
        t = torch.zeros(4, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)          # the zero fill is complete before any kernel can add to the buffer
            check(lib(f).rv_numeric_status_bind(C.c_void_p(t.data_ptr())), "rv_numeric_status_bind")
        _numeric[key] = t
    return t


def last_error():
    """The calling thread's last error message (of whichever loaded library has one)."""
    buf = C.create_string_buffer(512)
    msgs = []
    for h in _libs.values():
        h.rv_last_error(buf, 512)
        if buf.value:
            msgs.append(buf.value.decode())
    return " | ".join(msgs)


def check(rc, what):
    if rc != 0:
        # the error string is per library and thread and is not cleared by later successes: with both builds loaded, prefer the message that names
        # the entry point that just failed
        msgs = last_error().split(" | ")
        name = what.split("(")[0]
        mine = [m for m in msgs if name in m]
        raise HipLibraryError(f"{what} failed (status {rc}): {' | '.join(mine or msgs)}")


OPTION_KEYS = ("gemm_tile_variant", "gemm_cus", "gemm_arows", "fp8_decode", "fp8_prefill", "sample_variant", "rows_fill", "rows_spread", "rows_persistent", "rows_single", "gemm_waves", "gemm_mhalf", "precision", "lm_head_split", "last_block_rows", "adapter_stream16", "adapter_fold_t2v", "attn_lds", "qkv_lds")


class Options:
    """Owner of an OPTIONS-ONLY ``rv_ctx`` (``rv_ctx_create(NULL, ...)``): tunables for the building-block entry points that take
    an optional context (``rv_gemm``, ``rv_gemm_fp8``, ``rv_sample``).  ``Options(gemm_tile_variant=6)``; pass it as ``ctx=``.
    A context belongs to ONE library: ``flavour`` (default: the process default) names it."""

    def __init__(self, flavour=None, **kw):
        self.flavour = flavour_of(flavour)
        self.lib = lib(self.flavour)
        self._ctx = C.c_void_p()
        check(self.lib.rv_ctx_create(None, C.byref(self._ctx)), "rv_ctx_create(options)")
        for k, v in kw.items():
            self.set(k, v)

    def set(self, key, value):
        check(self.lib.rv_ctx_set_option(self._ctx, key.encode(), int(value)), f"rv_ctx_set_option({key})")
        return self

    def get(self, key):
        v = _i64()
        check(self.lib.rv_ctx_get_option(self._ctx, key.encode(), C.byref(v)), f"rv_ctx_get_option({key})")
        return int(v.value)

    def __del__(self):
        try:
            if self._ctx and self._ctx.value:
                self.lib.rv_ctx_destroy(self._ctx)
                self._ctx = C.c_void_p()
        except Exception:
            pass


def ctx_ptr(ctx, f=None):
    """``rv_ctx*`` of an ``Options`` / ``Engine`` (or None -> NULL: the library's default tunables).  ``f``: the flavour of the library
    about to be called - a context of the other library is refused."""
    if ctx is None:
        return None
    if f is not None and ctx.flavour != flavour_of(f):
        raise HipLibraryError(f"a {ctx.flavour} context was passed to a {flavour_of(f)} call")
    return ctx._ctx


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
