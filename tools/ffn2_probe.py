"""The adapter's FFN-2 GEMM (M = 25700 rows, N = 768, K = 2048, f32 out + bias + residual) under the kernel choices rv_gemm has:
default (output-tiled ping-pong 256 x 192), ring kernel only, forced stream-K on row chunks.   python tools/ffn2_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops  # noqa: E402
from kbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
OP = hip.op_dtype()
M, N, K = 25700, 768, 2048
x = torch.randn(M, K, device=dev).to(OP)
w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(OP))
b = torch.randn(N, device=dev)
res = torch.randn(M, N, device=dev)
out = torch.empty(M, N, device=dev)
fl = 2.0 * M * N * K
for name, opt in (("default", hip.Options()), ("ring only (6)", hip.Options(gemm_tile_variant=6)), ("pp tiled (4)", hip.Options(gemm_tile_variant=4))):
    us = timeit(lambda: ops.gemm(x, w, bias=b, residual=res, out=out, w_packed=True, ctx=opt))
    print(f"{name:22s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
sk = hip.Options(gemm_tile_variant=5)
for chunk in (8192, 6464, 5248, 4352):
    def run():
        for r0 in range(0, M, chunk):
            r1 = min(M, r0 + chunk)
            ops.gemm(x[r0:r1], w, bias=b, residual=res[r0:r1], out=out[r0:r1], w_packed=True, ctx=sk)
    try:
        us = timeit(run)
        print(f"stream-K, chunks of {chunk:5d} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
    except Exception as e:  # noqa: BLE001
        print(f"stream-K, chunks of {chunk}: {e}")
# the f32 epilogue's share: the same GEMM with a 16-bit output and no residual
o16 = torch.empty(M, N, dtype=OP, device=dev)
us = timeit(lambda: ops.gemm(x, w, out=o16, w_packed=True))
print(f"{'16-bit out, no residual':22s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
