"""bf16 vs FP8 x FP8 prefill GEMM (persistent ping-pong kernel) at the recursion's shapes; weights rotate over 8 copies (cold)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")


def timeit(fn, n=40, warm=5):
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


M = int(sys.argv[1]) if len(sys.argv) > 1 else 1005
for name, N, K, act, od in (("qkv(no rope)", 12288, 4096, 0, torch.float32), ("o", 4096, 4096, 0, torch.float32),
                            ("gate/up", 22016, 4096, 2, torch.bfloat16), ("down", 4096, 11008, 0, torch.float32)):
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) * 0.02) for _ in range(4)]
    wp = [ops.pack_fragments(w.to(torch.bfloat16)) for w in ws]
    w8 = [ops.pack_fragments_fp8_prefill(w) for w in ws]
    del ws
    out = torch.empty(M, N // 2 if act == 2 else N, dtype=od, device=dev)
    a8, sa = ops.quant_rows_fp8(x)
    i = {"n": 0}

    def f16():
        ops.gemm(x, wp[i["n"] % 4], out=out, act=act, w_packed=True)
        i["n"] += 1

    def f8():
        ops.gemm_fp8(a8, sa, w8[i["n"] % 4][0], w8[i["n"] % 4][1], out=out, act=act)
        i["n"] += 1

    fl = 2.0 * M * N * K
    supported = True
    try:
        f8()
    except Exception as e:  # noqa: BLE001
        supported = False
        print(name, "fp8 unsupported:", str(e)[:100])
    t16 = timeit(f16)
    tq = timeit(lambda: ops.quant_rows_fp8(x))
    if supported:
        t8 = timeit(f8)
        print(f"{name:14s} M={M} N={N} K={K}: bf16 {t16:7.1f} us {fl / t16 / 1e6:6.0f} TF | fp8 {t8:7.1f} us {fl / t8 / 1e6:6.0f} TF | quant rows {tq:5.1f} us")
    else:
        print(f"{name:14s} M={M}: bf16 {t16:7.1f} us")
