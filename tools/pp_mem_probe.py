"""Where the ping-pong GEMM's memory side spends its time: the same launch with the activation rows cold / hot (lda = 0: every row reads row 0)
in whatever build REVISION_HIP_LIB names (tools/pp_probe.sh: PP_ABL = 8 memory only, 16 weights hot, 24 both).  python tools/pp_mem_probe.py [M N K]"""
import sys
import torch
from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8040, 22016, 4096)
DT = hip.op_dtype()
lib = hip.lib()
x = (torch.randn(M, K, device=dev) * 0.5).to(DT)
ws_ = [ops.pack_fragments((torch.randn(N, K, device=dev) * 0.05).to(DT)) for _ in range(3)]
out = torch.empty(M, N, dtype=DT, device=dev)
ws = ops.stream_k_workspace(dev)
for lda in (K, 0):
    i = [0]

    def run():
        w = ws_[i[0] % 3]
        i[0] += 1
        hip.check(lib.rv_gemm(None, hip.ptr(x), lda, hip.ptr(w), K, 1, None, None, 0, hip.ptr(out), N, hip.dtype_code(out), 0, M, N, K,
                              hip.ptr(ws), ws.numel(), hip.stream()), "rv_gemm")
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 200 * 1e3
    tiles = ((M + 255) // 256) * (N // 256)
    print(f"M={M} N={N} K={K} lda={lda}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s   {us * 256 / tiles / (K // 64) * 1e3:6.0f} ns per k-tile and CU", flush=True)
