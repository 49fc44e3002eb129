"""The four LLM prefill projections at small row counts (the stage-1 passes: 4 .. 8 x 72 rows, 4 x 327 rows; partial batches): auto plan vs forced
persistent stream-K (gemm_tile_variant 5) vs the 128x128 ring kernel (6).  python3 tools/smallm_sweep.py [M ...]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
D, F = 4096, 11008
Ms = [int(a) for a in sys.argv[1:]] or [288, 504, 576, 1005, 1308, 1962]
W = {}
for name, N, K in (("qkv", 3 * D, D), ("o", D, D), ("gateup", 2 * F, D), ("down", D, F)):
    W[name] = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
for M in Ms:
    line = []
    for name, N, K, act, od in (("qkv", 3 * D, D, 0, torch.float32), ("o", D, D, 0, torch.float32), ("gateup", 2 * F, D, 2, torch.bfloat16), ("down", D, F, 0, torch.float32)):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        res = torch.randn(M, N, device=dev) if name in ("o", "down") else None
        out = torch.empty(M, N // 2 if act == 2 else N, dtype=od, device=dev)
        t = {}
        for v in (2, 5, 6):
            opt = hip.Options(gemm_tile_variant=v)
            try:
                t[v] = timeit(lambda: ops.gemm(x, W[name], residual=res, out=out, act=act, w_packed=True, ctx=opt), iters=30)
            except Exception as e:   # noqa: BLE001
                t[v] = float("nan")
        line.append(f"{name} auto {t[2]:6.1f} sk {t[5]:6.1f} ring {t[6]:6.1f}")
    print(f"M={M:5d}  " + " | ".join(line), flush=True)
