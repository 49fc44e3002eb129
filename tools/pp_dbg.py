import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops
dev = torch.device("cuda:0"); lib = hip.lib()
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, act, odt) in [(256, 256, 128, 0, torch.float32), (1005, 4096, 4096, 0, torch.float32), (1005, 22016, 4096, 2, torch.bfloat16),
                            (1005, 12288, 4096, 0, torch.float32), (1005, 4096, 11008, 0, torch.float32)]:
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    wp = ops.pack_fragments((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16))
    out = []
    for dbg in (0, 1, 2, 3):
        lib.rv_pp_debug(dbg)
        lib.rv_set_gemm_tile_variant(5)
        out.append(timeit(lambda: ops.gemm(a, wp, w_packed=True, stream_k=True, out_dtype=odt, act=act)))
    lib.rv_pp_debug(0)
    lib.rv_set_gemm_tile_variant(4)
    dp = timeit(lambda: ops.gemm(a, wp, w_packed=True, stream_k=False, out_dtype=odt, act=act))
    print(f"M={M} N={N} K={K}: full {out[0]:.1f} | no-publish-stores {out[1]:.1f} | no-finalize {out[2]:.1f} | neither {out[3]:.1f} | DP {dp:.1f} us", flush=True)
