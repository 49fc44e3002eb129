"""Counter-based synthetic weight initialiser, bit-identical on host (numpy) and device (HIP).

There are no checkpoints on the build / GPU boxes, so benchmarks and parity tests use random-init
weights of the reference's shapes.  Relying on torch RNG streams would give different bits on CPU and
GPU; instead every element is a pure function of (tensor name, global seed, flat index):

    h   = splitmix64(index + key)                 key = fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)
    v   = int(h >> 40) - 2**23                    exact 24-bit signed integer
    w   = float32(float64(v) * step + base)       product exact in double, one rounding to fp32

which gives uniform(base - a, base + a) with step = a * 2**-23.  ``csrc/init.hip`` (rv_init_hash)
computes the same thing on the GPU, so a 7B-parameter model is filled in HBM in milliseconds and the
oracle can rebuild exactly the same tensors on the CPU.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for c in name.encode("utf-8"):
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def tensor_key(name: str, seed: int) -> int:
    return (fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def step_for(a: float) -> np.float32:
    return np.float32(float(a) * 2.0 ** -23)


def hash_uniform(n: int, key: int, a: float, base: float = 0.0, offset: int = 0, chunk: int = 1 << 18):
    """fp32 array of n values uniform in [base - a, base + a)."""
    out = np.empty(n, dtype=np.float32)
    step = step_for(a)
    b = np.float32(base)
    with np.errstate(over="ignore"):
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            idx = np.arange(offset + s, offset + e, dtype=np.uint64) + np.uint64(key)
            h = _splitmix64(idx)
            v = (h >> np.uint64(40)).astype(np.int64) - np.int64(1 << 23)
            out[s:e] = (v.astype(np.float64) * np.float64(step) + np.float64(b)).astype(np.float32)
    return out


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (values stay fp32)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def round_f16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> fp16 -> fp32, saturating at +-65504 like the kernels' conversions (values stay fp32)."""
    return np.clip(np.ascontiguousarray(x, dtype=np.float32), -65504.0, 65504.0).astype(np.float16).astype(np.float32)


def round_op(x, op):
    """Round to an operand type's grid: ``op`` False / None (no rounding), True or "bf16", "f16"."""
    if op is None or op is False:
        return x
    if op is True or op == "bf16":
        return round_bf16(x)
    if op == "f16":
        return round_f16(x)
    raise ValueError(f"unknown operand type {op!r}")


def amplitude_pieces(a, shape):
    """``a`` is a float (one amplitude for the whole tensor) or a list of (row_end, amplitude) pieces over the leading dimension
    (row_end None = to the end): -> [(first element, element count, amplitude)] over the flat index.  The pieces share ONE hash
    stream (the key of the tensor, indexed by the flat element), so host and device fill a piece with the stream offset by its
    first element."""
    n = int(np.prod(shape))
    if not isinstance(a, (list, tuple)):
        return [(0, n, float(a))]
    row = n // int(shape[0])
    out, r0 = [], 0
    for r1, amp in a:
        r1 = int(shape[0]) if r1 is None else int(r1)
        assert r0 < r1 <= shape[0], (a, shape)
        out.append((r0 * row, (r1 - r0) * row, float(amp)))
        r0 = r1
    assert r0 == shape[0], (a, shape)
    return out


def make_tensor(name, shape, seed, a, base=0.0, bf16=False):
    n = int(np.prod(shape))
    key = tensor_key(name, seed)
    w = np.empty(n, dtype=np.float32)
    for e0, cnt, amp in amplitude_pieces(a, shape):
        w[e0:e0 + cnt] = hash_uniform(cnt, key, amp, base, offset=e0)
    w = w.reshape(shape)
    return round_op(w, bf16)
