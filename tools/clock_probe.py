"""Keep one kind of launch running for ~12 s so that `rocm-smi --showclocks --showpower` can be sampled next to it:
    python tools/clock_probe.py gemm|gemv|gemm_random
gemm = 4096^3 ping-pong GEMM on constant-ish data, gemm_random = the same on random data, gemv = decode gate/up weight stream."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, ops
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "gemm"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
if kind.startswith("gemm"):
    n = 4096
    x = torch.randn(n, n, device=dev).to(torch.bfloat16)
    w = ops.pack_fragments((torch.randn(n, n, device=dev) * 0.02).to(torch.bfloat16))
    out = torch.empty(n, n, dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm(x, w, out=out, w_packed=True)
    work = 2.0 * n ** 3
    unit = "TFLOP/s"
else:
    ws = [ops.pack_fragments((torch.randn(22016, 4096, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(8)]
    x = torch.randn(7, 4096, device=dev).to(torch.bfloat16)
    out = torch.empty(7, 11008, dtype=torch.bfloat16, device=dev)
    st = {"i": 0}
    def f():
        ops.gemm(x, ws[st["i"] % 8], act=hip.RV_ACT_SILU_MUL, out=out, w_packed=True)
        st["i"] += 1
    work = 2.0 * 22016 * 4096
    unit = "GB/s"
for _ in range(10):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
n_it = 0
while time.perf_counter() - t0 < secs:
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    n_it += 200
dt = time.perf_counter() - t0
print(f"{kind}: {work * n_it / dt / (1e12 if unit == 'TFLOP/s' else 1e9):.0f} {unit} sustained over {dt:.1f} s")
