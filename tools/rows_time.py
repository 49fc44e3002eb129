"""Time the 33..144-row split-K decode kernel alone on synthetic operands: us per launch vs K and N (fit the fixed cost).
python tools/rows_time.py [M]"""
import ctypes
import sys
import torch
from revisionllm_amd import hip

lib = hip.lib()
f = lib.rv_gemm_rows
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 63
from revisionllm_amd import ops
mbp = ops.xp_blocks(M)
planes = torch.zeros(lib.rv_gemm_rows_ws_bytes(), dtype=torch.uint8, device=dev)
arrive = torch.zeros(4096, dtype=torch.int32, device=dev)
for N in (4096, 12288, 22016):
    for K in (1024, 2048, 4096, 8192, 16384):
        nw = max(2, int(8e8 // (N * K * 2)))
        ws = [torch.randn(N * K // 2, device=dev).view(torch.int32) for _ in range(nw)]       # any bits: timing only
        x = (torch.randn(mbp * 16 * K, device=dev) * 0.1).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev)
        i = [0]

        def run():
            rc = f(hip.ptr(x), hip.ptr(ws[i[0] % nw]), None, hip.ptr(c), M, N, K, hip.ptr(planes), hip.ptr(arrive), hip.RV_ACT_NONE, hip.RV_F32, hip.stream())
            assert rc == 0, hip.last_error()
            i[0] += 1
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 40 * 1e3
        print(f"M={M} N={N:5d} K={K:5d}: {us:7.1f} us  {2.0 * N * K / us / 1e3:6.0f} GB/s", flush=True)
