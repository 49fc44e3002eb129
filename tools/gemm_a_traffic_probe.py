"""How much of the prefill gate/up GEMM's time is the activation panel's re-reads?  Same launch with lda = 0 (every row reads row 0:
the A operand stays in L2) against the real row stride.  python tools/gemm_a_traffic_probe.py [M]"""
import sys
import torch
from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4020
lib = hip.lib()
for name, N, K, act, od in (("gateup", 22016, 4096, hip.RV_ACT_SILU_MUL, torch.bfloat16), ("down", 4096, 11008, 0, torch.float32), ("qkv", 12288, 4096, 0, torch.float32)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
    out = torch.empty(M, N // 2 if act == 2 else N, dtype=od, device=dev)
    ws = ops.stream_k_workspace(dev)
    for lda in (K, 0):
        def run():
            hip.check(lib.rv_gemm(None, hip.ptr(x), lda, hip.ptr(w), K, 1, None, None, 0, hip.ptr(out), out.shape[1], hip.dtype_code(out), act, M, N, K,
                                  hip.ptr(ws), ws.numel(), hip.stream()), "rv_gemm")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print(f"{name} M={M} lda={lda}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
