"""Micro-benchmarks of the individual HIP kernels on the shapes the stage-2 recursion launches (HIP events on the
launch stream).  Usage (GPU box): python tools/kbench.py [gemm] [gemv] [attn] [small]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
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
    which = set(sys.argv[1:]) or {"gemm", "gemv", "attn"}
    dev = torch.device("cuda:0")
    D, F, V = 4096, 11008, 32000
    for geo in ((2, 3) if "gemm" in which else ()):
        opt = hip.Options(gemm_tile_variant=geo)
        print(f"--- tile variant {geo}")
        for M in (1005, 4020):
            for name, N, K, act, od in (("qkv", 3 * D, D, 0, torch.float32), ("o", D, D, 0, torch.float32),
                                        ("gateup", 2 * F, D, 2, torch.bfloat16), ("down", D, F, 0, torch.float32)):
                x = torch.randn(M, K, device=dev).to(torch.bfloat16)
                w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
                res = torch.randn(M, N, device=dev) if name in ("o", "down") else None
                out = torch.empty(M, N // 2 if act == 2 else N, dtype=od, device=dev)
                us = timeit(lambda: ops.gemm(x, w, residual=res, out=out, act=act, w_packed=True, ctx=opt))
                us0 = timeit(lambda: ops.gemm(x, w, residual=res, out=out, act=act, w_packed=True, stream_k=False, ctx=opt))
                print(f"gemm {name:7s} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s   (tiled: {us0:7.1f} us)")
        for name, M, N, K, act in (("dense.proj", 25600, 4096, 768, 0), ("adp.qk", 25700, 1536, 768, 0), ("adp.v", 25700, 768, 768, 0), ("adp.ffn1", 25700, 2048, 768, 1),
                                   ("adp.ffn2", 25700, 768, 2048, 0)):
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            us = timeit(lambda: ops.gemm(x, w, out=out, act=act, w_packed=True, ctx=opt))
            print(f"gemm {name:8s} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")
    if "gemv" in which:
        for M in (7, 16, 21, 28):
            for name, N, K, act, od in (("qkv", 3 * D, D, 0, torch.float32), ("o", D, D, 0, torch.float32),
                                        ("gateup", 2 * F, D, 2, torch.bfloat16), ("down", D, F, 0, torch.float32),
                                        ("lm_head", V, D, 0, torch.float32)):
                x = torch.randn(M, K, device=dev).to(torch.bfloat16)
                ws = [ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(max(2, int(6e8 // (N * K * 2))))]
                out = torch.empty(M, N // 2 if act == 2 else N, dtype=od, device=dev)
                st = {"i": 0}

                def f():
                    ops.gemm(x, ws[st["i"] % len(ws)], out=out, act=act, w_packed=True)
                    st["i"] += 1
                us = timeit(f, iters=40, warm=4)
                w8s = [ops.pack_fragments_fp8((torch.randn(N, K, device=dev) * 0.02)) for _ in range(max(2, int(6e8 // (N * K))))]

                def f8():
                    w8, sc = w8s[st["i"] % len(w8s)]
                    ops.gemv_fp8(x, w8, sc, out=out, act=act)
                    st["i"] += 1
                us8 = timeit(f8, iters=40, warm=4)
                print(f"gemv {name:7s} M={M:2d} N={N} K={K}: {us:7.1f} us  {2.0*N*K/us/1e3:7.0f} GB/s | fp8 weights {us8:7.1f} us {1.0*N*K/us8/1e3:7.0f} GB/s")
    if "attn" in which:
        for B, Lq, Lk, H, dh, causal, q0 in ((7, 151, 151, 32, 128, True, 0), (7, 1, 158, 32, 128, True, 157), (100, 257, 257, 8, 96, False, 0)):
            q = torch.randn(B, Lq, H, dh, device=dev).to(torch.bfloat16)
            k = torch.randn(B, Lk, H, dh, device=dev).to(torch.bfloat16)
            v = torch.randn(B, Lk, H, dh, device=dev).to(torch.bfloat16)
            us = timeit(lambda: ops.attention(q, k, v, causal=causal, q_pos0=q0))
            print(f"attn(+transposes in wrapper) B={B} Lq={Lq} Lk={Lk} H={H} dh={dh}: {us:8.1f} us")


if __name__ == "__main__":
    main()
