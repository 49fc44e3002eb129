"""Yardstick only (never on the product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the prefill GEMM shapes, next to
rv_gemm on the same shapes, in the operand type of the library build under test (fp16 by default; `python tools/blas_yardstick.py bf16`).  Both sides get a 16-bit
output and no bias / residual: the main loops are compared, not the fused epilogues the engine runs.  Usage: python tools/blas_yardstick.py [f16|bf16]"""
import sys
import torch

import os

from revisionllm_amd import hip, ops

if len(sys.argv) > 1:
    hip.set_flavour(sys.argv[1])
DT = hip.op_dtype()
print(f"operands: {hip.flavour()} ({DT}); torch {torch.__version__}", flush=True)
dev = torch.device("cuda:0")
OPT = hip.Options(gemm_waves=int(os.environ["WAVES"])) if os.environ.get("WAVES") else None
if len(sys.argv) > 1 and OPT is not None:
    OPT = hip.Options(flavour=sys.argv[1], gemm_waves=int(os.environ["WAVES"]))   # WAVES=4 / 8: the rv_gemm form
shapes = [(4020, 22016, 4096), (4020, 4096, 4096), (4020, 4096, 11008), (4020, 12288, 4096), (2010, 22016, 4096), (4096, 4096, 4096), (8192, 8192, 8192)]
if os.environ.get("ROWS"):                       # ROWS=8040: the four projection shapes of a pass of that many rows only
    shapes = [(int(os.environ["ROWS"]), n, k) for n, k in ((22016, 4096), (4096, 4096), (4096, 11008), (12288, 4096))]
if os.environ.get("KSWEEP"):                     # KSWEEP=1: N = 4096 at growing K - a line fit separates the per-k-tile time from the per-tile fixed cost
    shapes = [(int(os.environ.get("ROWS", 8040)), 4096, k) for k in (512, 1024, 2048, 4096, 8192, 16384)]


def clocks_of(fn, seconds=1.5):
    """CLOCKS=1: mean shader clock / socket power over `seconds` of back-to-back launches (bench.ClockSampler: sysfs, no GPU call)."""
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import ClockSampler
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    with ClockSampler(dev, period=0.02) as cs:
        while time.time() - t0 < seconds:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
    s = cs.summary()
    return f"[{s['sclk_mhz_mean']:.0f} MHz {s['power_w_mean']:.0f} W]" if s.get("available") else "[clocks n/a]"


def timeit(fn, n=20, warm=5, min_ms=300.0):
    """Sustained rate: warm up for min_ms / 2, then time at least min_ms of back-to-back launches (the clock ramps over milliseconds)."""
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        fn()
    b.record()
    torch.cuda.synchronize()
    one = max(a.elapsed_time(b) / 4, 1e-3)
    n = max(n, int(min_ms / one))
    warm = max(warm, n // 2)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for M, N, K in shapes:
    copies = 6                                   # rotate weights: the 256 MB infinity cache must not serve them
    x = (torch.randn(M, K, device=dev) * 0.5).to(DT)
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(DT) for _ in range(copies)]
    i = [0]

    def blas():
        torch.matmul(x, ws[i[0] % copies].t())
        i[0] += 1
    t_blas = timeit(blas)
    wps = [ops.pack_fragments(w) for w in ws]
    line = f"M={M:5d} N={N:5d} K={K:5d}  torch.matmul {t_blas * 1e3:7.1f} us  {2.0 * M * N * K / t_blas / 1e9:7.1f} TF/s"
    if os.environ.get("CLOCKS"):
        line += " " + clocks_of(blas)
    if wps is not None:
        out = torch.empty(M, N, dtype=DT, device=dev)

        def mine():
            ops.gemm(x, wps[i[0] % copies], out=out, w_packed=True, ctx=OPT)
            i[0] += 1
        try:
            t = timeit(mine)
            line += f"   rv_gemm {t * 1e3:7.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TF/s"
            if os.environ.get("CLOCKS"):
                line += " " + clocks_of(mine)
        except Exception as e:  # noqa: BLE001
            line += f"   rv_gemm failed: {e}"
    print(line, flush=True)
