"""Average launch time of rv_sample on the decode shape ([7, 32000] logits, T=0.05, top_k=50)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(B, 32000, generator=g) * 2.0).to(dev)
    u = torch.rand(B, generator=g).to(dev)
    from revisionllm_amd import hip
    o = ops.sample(logits, u, True, 0.05, 50, 1.0)
    lib, st = hip.lib(), hip.stream()
    args = (None, hip.ptr(logits), B, 32000, hip.ptr(u), 1, 0.05, 50, 1.0, hip.ptr(o["tokens"]), hip.ptr(o["entropy_proc"]),
            hip.ptr(o["entropy_raw"]), hip.ptr(o["topk_idx"]), hip.ptr(o["topk_val"]), hip.ptr(o["n_keep"]), hip.ptr(o["threshold"]), st)
    for _ in range(5):
        lib.rv_sample(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(200):
        lib.rv_sample(*args)
    b.record()
    torch.cuda.synchronize()
    print(f"rv_sample B={B}: {a.elapsed_time(b) * 5:.1f} us per launch (back to back, outputs preallocated); tokens {o['tokens'].tolist()}")
    if os.environ.get("RV_SAMPLE_PROBE"):
        print("phase ticks (10 ns): load+entropy, radix, compaction+rank, serial tail:", o["topk_val"][:2, 60:64].tolist(), "since phase 1: selectA, count, compaction, selectB", o["topk_val"][:2, 55:59].tolist())
    print("entropy_raw", o["entropy_raw"].tolist())
    # output fingerprint over several settings (compare builds: REVISION_HIP_LIB=...)
    import hashlib
    h = hashlib.sha256()
    lt = (logits * 2).round() / 2
    for x in (logits, lt, logits * 0.01):
        for (temp, k, p) in ((0.05, 50, 1.0), (0.05, 50, 0.6), (1.0, 64, 0.9), (0.7, 7, 1.0)):
            o = ops.sample(x, u, True, temp, k, p)
            for name in ("tokens", "entropy_proc", "entropy_raw", "topk_idx", "topk_val", "n_keep"):
                h.update(o[name].cpu().numpy().tobytes())
    print("fingerprint", h.hexdigest()[:16])


if __name__ == "__main__":
    main()
