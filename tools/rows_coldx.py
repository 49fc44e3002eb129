"""Does the wide decode kernel pay for COLD activations?  Inside a decode step the activation operand x (144 rows x K bf16, ~1.2 MB) was
written by the previous kernel an instant ago, from other XCDs: every XCD's L2 misses on its first touch of every line.  The stand-alone
timings (tools/rows_time.py, the bench's roofline leg) re-read ONE x that all eight L2s hold after the first launch.
Three timings per decode shape at M rows: x hot (one buffer), x rotated over 64 buffers (L2-cold, infinity-cache-warm at best), and x
re-written by a copy kernel right before every launch (what a decode step does; the copy's own time is measured and subtracted).
    python tools/rows_coldx.py [M]"""
import sys

import torch

from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 140
lib = hip.lib()
mbp = ops.xp_blocks(M)
planes = torch.zeros(lib.rv_gemm_rows_ws_bytes(), dtype=torch.uint8, device=dev)
arrive = torch.zeros(4096, dtype=torch.int32, device=dev)


def ev(fn, n=64, warm=8):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, N, K, act, odt in (("qkv", 12288, 4096, hip.RV_ACT_NONE, hip.RV_F32), ("o", 4096, 4096, hip.RV_ACT_NONE, hip.RV_F32),
                             ("gate/up", 22016, 4096, hip.RV_ACT_SILU_MUL, hip.RV_BF16), ("down", 4096, 11008, hip.RV_ACT_NONE, hip.RV_F32)):
    nw = max(3, int(6e8 // (N * K * 2)) + 1)
    ws = [torch.randn(N * K // 2, device=dev).view(torch.int32) for _ in range(nw)]
    xs = [(torch.randn(mbp * 16 * K, device=dev) * 0.1).to(torch.bfloat16) for _ in range(64)]
    src = (torch.randn(mbp * 16 * K, device=dev) * 0.1).to(torch.bfloat16)
    c = torch.empty(M * N, device=dev)
    i = [0]

    def run(x):
        rc = lib.rv_gemm_rows(hip.ptr(x), hip.ptr(ws[i[0] % nw]), None, hip.ptr(c), M, N, K, hip.ptr(planes), hip.ptr(arrive), act, odt, hip.stream())
        assert rc == 0, hip.last_error()
        i[0] += 1
    hot = ev(lambda: run(xs[0]))
    rot = ev(lambda: run(xs[i[0] % 64]))
    cp = ev(lambda: xs[0].copy_(src))

    def fresh():
        xs[0].copy_(src)
        run(xs[0])
    fr = ev(fresh)
    print(f"M={M} {name:8s} N={N:5d} K={K:5d}: x hot {hot:6.1f} us | x rotated over 64 buffers {rot:6.1f} us | x re-written before every launch {fr - cp:6.1f} us "
          f"(copy alone {cp:4.1f} us)", flush=True)
    del ws, xs
