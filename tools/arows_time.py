"""A/B timing of the short-K many-row GEMM family: A-resident kernel (gemm_arows=1) vs the 128x128 ring kernel (0).
HIP events on the launch stream.  python tools/arows_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    on, off = hip.Options(gemm_arows=1), hip.Options(gemm_arows=0)
    for name, M, N, K, act, od in (("dense.proj", 25600, 4096, 768, 0, torch.bfloat16), ("adp.qk", 25700, 1536, 768, 0, torch.bfloat16),
                                   ("adp.v", 25700, 768, 768, 0, torch.bfloat16), ("adp.out", 25700, 768, 768, 0, torch.float32),
                                   ("adp.ffn1", 25700, 2048, 768, 1, torch.bfloat16), ("adp.q.t2v", 25600, 768, 768, 0, torch.bfloat16),
                                   ("s33.proj", 8448, 4096, 768, 0, torch.bfloat16), ("s33.qk", 8481, 1536, 768, 0, torch.bfloat16), ("s16.ffn1", 4112, 2048, 768, 1, torch.bfloat16), ("s24.qk", 6168, 1536, 768, 0, torch.bfloat16), ("sparse.qk", 102500, 1536, 768, 0, torch.bfloat16)):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if od == torch.float32 else None
        out = torch.empty(M, N, dtype=od, device=dev)
        t1 = timeit(lambda: ops.gemm(x, w, bias=bias, residual=res, out=out, act=act, w_packed=True, stream_k=False, ctx=on))
        t0 = timeit(lambda: ops.gemm(x, w, bias=bias, residual=res, out=out, act=act, w_packed=True, stream_k=False, ctx=off))
        fl = 2.0 * M * N * K
        by = M * K * 2 + M * N * out.element_size() + (M * N * 4 if res is not None else 0)
        print(f"{name:10s} M={M:6d} N={N:4d} K={K:4d}: A-resident {t1:7.1f} us {fl/t1/1e6:6.0f} TF/s {by/t1/1e3:6.0f} GB/s | ring {t0:7.1f} us {fl/t0/1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
