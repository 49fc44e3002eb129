"""Yardstick only (never on the product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the DECODE projection shapes of a merged
step with 70 / 112 / 140 / 144 rows, next to rv_gemm_rows (the wide decode kernel) and rv_gemm (the prefill tile kernels) on the same
shapes.  Weights are rotated over > 256 MB so the infinity cache does not serve them.
Usage: python tools/blas_yardstick_m144.py [rows ...]"""
import sys

import torch

from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")
ROWS = [int(a) for a in sys.argv[1:]] or [144, 140, 112, 70]
SHAPES = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate/up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)]


def timeit(fn, min_ms=200.0):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        fn()
    b.record()
    torch.cuda.synchronize()
    one = max(a.elapsed_time(b) / 4, 1e-3)
    n = max(20, int(min_ms / one))
    for _ in range(n // 2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for M in ROWS:
    for name, N, K in SHAPES:
        copies = max(3, int(6e8 // (N * K * 2)) + 1)
        x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(copies)]
        i = [0]
        out_b = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

        def blas():
            torch.matmul(x, ws[i[0] % copies].t(), out=out_b)
            i[0] += 1
        t_blas = timeit(blas)
        gb = N * K * 2 / 1e9
        line = f"M={M:3d} {name:8s} N={N:5d} K={K:5d}  torch.matmul {t_blas * 1e3:6.1f} us {gb / t_blas:6.2f} TB/s"
        wps = [ops.pack_fragments(w) for w in ws]
        del ws
        xp = ops.pack_rows(x)
        out = torch.empty(M, N, dtype=torch.float32, device=dev)

        def rows():
            ops.gemm_rows(x, wps[i[0] % copies], out=out, xp=xp)
            i[0] += 1
        try:
            t = timeit(rows)
            line += f"   rv_gemm_rows {t * 1e3:6.1f} us {gb / t:6.2f} TB/s"
        except Exception as e:  # noqa: BLE001
            line += f"   rv_gemm_rows failed: {str(e)[:60]}"
        out16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

        def tiles():
            ops.gemm(x, wps[i[0] % copies], out=out16, w_packed=True)
            i[0] += 1
        try:
            t = timeit(tiles)
            line += f"   rv_gemm {t * 1e3:6.1f} us {gb / t:6.2f} TB/s"
        except Exception as e:  # noqa: BLE001
            line += f"   rv_gemm failed: {str(e)[:60]}"
        print(line, flush=True)
        del wps
