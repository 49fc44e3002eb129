"""Is the lda = 0 speed-up of tools/gemm_a_traffic_probe.py memory traffic or POWER?  Same gate/up launch at M rows with (a) random rows,
(b) lda = 0 (one L2-hot row, identical operands in every m-tile), (c) the real row stride but every row a copy of row 0 (full traffic from
the infinity cache, identical operand bits).  python tools/gemm_a_power_probe.py [M]"""
import sys
import torch
from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4020
lib = hip.lib()
N, K, act = 22016, 4096, hip.RV_ACT_SILU_MUL
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
xsame = x[:1].repeat(M, 1).contiguous()
w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
ws = ops.stream_k_workspace(dev)
for name, a, lda in (("random rows, lda = K", x, K), ("lda = 0 (row 0 for every row)", x, 0), ("identical rows, lda = K", xsame, K), ("random rows again", x, K)):
    def run():
        hip.check(lib.rv_gemm(None, hip.ptr(a), lda, hip.ptr(w), K, 1, None, None, 0, hip.ptr(out), out.shape[1], hip.dtype_code(out), act, M, N, K,
                              hip.ptr(ws), ws.numel(), hip.stream()), "rv_gemm")
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 60 * 1e3
    print(f"gate/up M={M} {name:32s}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
