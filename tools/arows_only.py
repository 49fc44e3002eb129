"""Launch the dense-projector GEMM (25600 x 4096 x 768) a few times with one kernel family (for rocprofv3 --pmc runs).
python tools/arows_only.py [gemm_arows option: 0 ring, 1 auto, 2 / 4 forced slice width] [M] [N] [K]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops
dev = torch.device("cuda:0")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M, N, K = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 25600), (3, 4096), (4, 768)))
opt = hip.Options(gemm_arows=mode)
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    ops.gemm(x, w, bias=bias, out=out, w_packed=True, stream_k=False, ctx=opt)
torch.cuda.synchronize()
