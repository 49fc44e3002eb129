import math, torch
from revisionllm_amd import hip, ops
dev="cuda:0"
for (M,N,K,act) in [(70,4096,4096,0),(70,4096,11008,0)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    w8, sc = ops.pack_fragments_fp8(w)
    q, _ = ops.quantize_rows_fp8(w)
    wd = q.double() * sc.double()[:, None]
    y = ops.gemm_rows(x, w8, act=act, out_dtype=torch.float32, w_scale=sc)
    z = x.double() @ wd.t()
    lo = ops.gemv_fp8(x[:16], w8, sc, out_dtype=torch.float32, act=act)
    print(M,N,K, "rows vs f64", float((y.double()-z).abs().max()/z.abs().max()), "gemv16 vs f64", float((lo.double()-z[:16]).abs().max()/z.abs().max()),
          "equal", torch.equal(y[:16], lo))
    print(y[0,:8].tolist(), z[0,:8].tolist())
    # which k contributes? zero all but first 128 k
    for kk in (128, 256, 1024):
        x2 = x.clone(); x2[:, kk:] = 0
        y2 = ops.gemm_rows(x2, w8, act=act, out_dtype=torch.float32, w_scale=sc)
        z2 = x2.double() @ wd.t()
        print("  k<%d" % kk, float((y2.double()-z2).abs().max()/z2.abs().max()))
