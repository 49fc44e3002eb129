"""Skinny decode-shaped GEMMs (M = 28 .. 128 rows, weights rotated so the infinity cache cannot serve them): us per launch and
the weight-streaming rate, for whatever kernel rv_gemm picks.  python tools/skinny_time.py [M ...]"""
import sys
import torch
from revisionllm_amd import hip, ops


def timeit(fn, iters=40, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


dev = torch.device("cuda:0")
D, F = 4096, 11008
Ms = [int(a) for a in sys.argv[1:]] or [28, 56, 64, 112, 128]
for M in Ms:
    tot = 0.0
    for name, N, K, act, od in (("qkv", 3 * D, D, 0, torch.float32), ("o", D, D, 0, torch.float32), ("gateup", 2 * F, D, 2, torch.bfloat16), ("down", D, F, 0, torch.float32)):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        ws = [ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(max(2, int(6e8 // (N * K * 2))))]
        res = torch.randn(M, N, device=dev) if name in ("o", "down") else None
        out = torch.empty(M, N // 2 if act == 2 else N, dtype=od, device=dev)
        st = {"i": 0}

        def f():
            ops.gemm(x, ws[st["i"] % len(ws)], residual=res, out=out, act=act, w_packed=True)
            st["i"] += 1
        us = timeit(f)
        tot += us
        print(f"M={M:3d} {name:7s} N={N} K={K}: {us:7.1f} us  {2.0*N*K/us/1e3:7.0f} GB/s  {2.0*M*N*K/us/1e6:6.1f} TF/s", flush=True)
    print(f"M={M:3d} layer total {tot:7.1f} us -> {tot*32/1e3:.2f} ms / step, {tot*32/1e3/M*7:.3f} ms per 7-row generate", flush=True)
