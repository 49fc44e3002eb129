"""Launch time of rv_attention on the recursion's two attention shapes (KV-cache layout: K [B,H,Smax,dh], V^T [B,H,dh,Smax])."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip

dev = torch.device("cuda:0")
lib, st = hip.lib(), hip.stream()


def run(name, B, H, dh, Lq, Lk, Smax, causal, q_pos0):
    D = H * dh
    torch.manual_seed(0)
    q = (torch.randn(B, Lq, D, device=dev) * 0.5).to(torch.bfloat16)
    k = (torch.randn(B, H, Smax, dh, device=dev) * 0.5).to(torch.bfloat16)
    vt = (torch.randn(B, H, dh, Smax, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.empty(B, Lq, D, dtype=torch.bfloat16, device=dev)
    args = (hip.ptr(q), D, Lq * D, hip.ptr(k), dh, H * Smax * dh, Smax * dh, hip.ptr(vt), H * dh * Smax, dh * Smax, Smax, hip.ptr(out), D,
            Lq * D, None, B, H, dh, Lq, Lk, int(causal), q_pos0, 1, 1.0 / math.sqrt(dh), st)
    for _ in range(5):
        hip.check(lib.rv_attention(*args), "rv_attention")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(100):
        lib.rv_attention(*args)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 10
    fl = 4.0 * B * H * Lq * Lk * dh * (0.5 if causal and q_pos0 == 0 else 1.0)
    import hashlib
    print(f"{name}: {us:7.1f} us  ({fl / us / 1e6:6.1f} TF/s nominal)  out sha {hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:10]}")


run("prefill rows  B=7  H=32 dh=128 Lq=139 Lk=171 causal", 7, 32, 128, 139, 171, 192, True, 32)
run("prefill full  B=7  H=32 dh=128 Lq=171 Lk=171 causal", 7, 32, 128, 171, 171, 192, True, 0)
run("adapter       B=100 H=8 dh=96  Lq=257 Lk=257       ", 100, 8, 96, 257, 257, 288, False, 0)
run("decode        B=7  H=32 dh=128 Lq=1   Lk=176       ", 7, 32, 128, 1, 176, 192, False, 175)
if len(sys.argv) > 1:   # scaling probes (latency floor vs throughput: B=1 12.7 us, B=7 24.6, B=28 73)
    run("probe B=1 Lq=139 Lk=171", 1, 32, 128, 139, 171, 192, True, 32)
    run("probe B=7 Lq=139 Lk=171 non-causal", 7, 32, 128, 139, 171, 192, False, 0)
    run("probe B=7 Lq=64 Lk=64 causal", 7, 32, 128, 64, 64, 192, True, 0)
    run("probe B=7 Lq=139 Lk=32", 7, 32, 128, 139, 32, 192, False, 0)
    run("probe B=28 Lq=139 Lk=171 causal", 28, 32, 128, 139, 171, 192, True, 32)
