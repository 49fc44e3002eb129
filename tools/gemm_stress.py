import sys, math, torch
sys.path.insert(0, "tests")
from helpers import feats, rel_err


def bf(x):
    return x.to(torch.bfloat16)
from revisionllm_amd import ops
dev = torch.device("cuda:0")
shapes = [(300, 768, 768), (130, 512, 1408), (1190, 4096, 512), (257, 2304, 768), (100, 4096, 768), (5000, 1536, 768)]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for (M, N, K) in shapes:
        a = feats(f"gemm.a.{M}.{K}", (M, K), bf16=True)
        w = bf(feats(f"gemm.w.{N}.{K}", (N, K), bf16=True) * (1.0 / math.sqrt(K))).float()
        ad, wd = bf(a).to(dev), bf(w).to(dev)
        ref0 = a.double() @ w.double().t()
        for od in (torch.bfloat16, torch.float32):
            y = ops.gemm(ad, wd, out_dtype=od)
            e = rel_err(y.float().cpu(), ref0)
            if e > (0.008 if od == torch.bfloat16 else 1e-4):
                bad += 1
                d = (y.float().cpu().double() - ref0).abs()
                idx = (d > 0.5 * d.max()).nonzero()
                print("BAD", it, M, N, K, od, e, "n_bad_elems", int((d > 0.004 * ref0.abs().max()).sum()), idx[:8].tolist(), flush=True)
print("done bad =", bad)
