"""A/B of the 256x256x64 ping-pong GEMM (tile variant 4) against the 128x128 ring kernel (variant 2): bitwise equality
over repeated launches (race screen) and timing at the path's shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")
lib = hip.lib()
OPTS = {v: hip.Options(gemm_tile_variant=v, gemm_arows=0) for v in (2, 4, 5, 6)}


def run(variant, a, wp, **kw):
    return ops.gemm(a, wp, w_packed=True, stream_k=(variant == 5), ctx=OPTS[variant], **kw)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
bad = 0
for (M, N, K, act, odt, res) in [(256, 256, 64, 0, torch.float32, False), (256, 256, 128, 0, torch.float32, False),
                                 (256, 512, 256, 0, torch.float32, False), (1005, 4096, 4096, 0, torch.float32, True), (171, 4096, 4096, 0, torch.float32, True), (400, 12288, 4096, 0, torch.float32, False), (700, 4096, 11008, 0, torch.float32, True),
                                 (1005, 22016, 4096, 2, torch.bfloat16, False), (300, 768, 2048, 1, torch.bfloat16, False),
                                 (1005, 12288, 4096, 0, torch.float32, False), (1197, 4096, 11008, 0, torch.float32, True), (2010, 22016, 4096, 2, torch.bfloat16, False), (2010, 4096, 11008, 0, torch.float32, True), (2010, 12288, 4096, 0, torch.float32, False), (700, 22016, 4096, 2, torch.bfloat16, False),
                                 (25700, 1536, 768, 0, torch.bfloat16, False)]:
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    wp = ops.pack_fragments(w)
    bias = torch.randn(N, device=dev) if act != 2 else None
    r = torch.randn(M, N, device=dev) if res else None
    ref = run(2, a, wp, bias=bias, residual=r, out_dtype=odt, act=act)
    nbad = 0
    for rep in range(8):
        out = run(4, a, wp, bias=bias, residual=r, out_dtype=odt, act=act)
        if not torch.equal(out, ref):
            nbad += 1
            d = (out.float() - ref.float()).abs()
            print(f"   rep {rep}: mismatch max {d.max().item():.4g} at {(d > 0).sum().item()} elements")
    bad += nbad
    sk_err, sk_bad = 0.0, 0
    first = None
    for rep in range(8):
        out = run(5, a, wp, bias=bias, residual=r, out_dtype=odt, act=act)
        if first is None:
            first = out.clone()
            sk_err = ((out.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
        elif not torch.equal(out, first):
            sk_bad += 1
    bad += sk_bad
    t2 = timeit(lambda: run(2, a, wp, bias=bias, residual=r, out_dtype=odt, act=act))
    t4 = timeit(lambda: run(4, a, wp, bias=bias, residual=r, out_dtype=odt, act=act))
    t5 = timeit(lambda: run(5, a, wp, bias=bias, residual=r, out_dtype=odt, act=act))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} act={act}: {'OK ' if nbad == 0 else 'BAD'}  ring {t2:7.1f} us {fl / t2 / 1e6:6.0f} TF | pingpong {t4:7.1f} us {fl / t4 / 1e6:6.0f} TF"
          f" | stream-K {t5:7.1f} us {fl / t5 / 1e6:6.0f} TF  (rel err vs ring {sk_err:.2e}, nondeterministic reps {sk_bad})")
for n in (4096, 8192):
    a = (torch.randn(n, n, device=dev) * 0.5).to(torch.bfloat16)
    wp = ops.pack_fragments((torch.randn(n, n, device=dev) * 0.05).to(torch.bfloat16))
    t2 = timeit(lambda: run(2, a, wp, out_dtype=torch.bfloat16), n=10)
    t4 = timeit(lambda: run(4, a, wp, out_dtype=torch.bfloat16), n=10)
    print(f"{n}^3: ring {2.0 * n ** 3 / t2 / 1e6:6.0f} TF | pingpong {2.0 * n ** 3 / t4 / 1e6:6.0f} TF")
print("mismatching launches:", bad)
