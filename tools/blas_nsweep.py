"""Yardstick: N sweep at M = 4020, K = 4096 (torch.matmul vs rv_gemm) - is the gate/up shape (N = 22016) slow because of its tile count?"""
import sys
import torch
from revisionllm_amd import ops

dev = torch.device("cuda:0")
M, K = int(sys.argv[1]) if len(sys.argv) > 1 else 4020, 4096


def timeit(fn, n=20, warm=5):
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


x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
for N in (8192, 12288, 16384, 19456, 20480, 21504, 22016, 22528, 24576, 32768):
    copies = 4
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(copies)]
    wps = [ops.pack_fragments(w) for w in ws]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    i = [0]

    def blas():
        torch.matmul(x, ws[i[0] % copies].t(), out=out)
        i[0] += 1

    def mine():
        ops.gemm(x, wps[i[0] % copies], out=out, w_packed=True)
        i[0] += 1
    tb, tm = timeit(blas), timeit(mine)
    fl = 2.0 * M * N * K
    print(f"N={N:6d} tiles={((M + 255) // 256) * (N // 256):5d} ({((M + 255) // 256) * (N // 256) / 256:.3f} rounds)  torch {tb * 1e3:7.1f} us {fl / tb / 1e9:7.1f} TF/s   rv_gemm {tm * 1e3:7.1f} us {fl / tm / 1e9:7.1f} TF/s", flush=True)
    del ws, wps
