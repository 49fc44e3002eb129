"""Launch the prefill GEMMs of one layer a few times (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1005
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # option gemm_tile_variant: 2 auto, 4 ping-pong tiled, 5 stream-K, 6 ring
opt = hip.Options(gemm_tile_variant=variant)
D, F = 4096, 11008
for name, N, K, act, od in (("qkv", 3 * D, D, 0, torch.float32), ("gateup", 2 * F, D, 2, torch.bfloat16), ("down", D, F, 0, torch.float32)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
    out = torch.empty(M, N // 2 if act == 2 else N, dtype=od, device=dev)
    for _ in range(3):
        ops.gemm(x, w, out=out, act=act, w_packed=True, stream_k=variant in (2, 5), ctx=opt)
torch.cuda.synchronize()
