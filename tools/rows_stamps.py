"""Cycle stamps of ONE consumer wave of the 9-row-block decode kernel (a library built with -DRS_PROBE=16384, gemm_rows.hip): where do a stage's cycles go?
    REVISION_HIP_LIB=<probe library> python tools/rows_stamps.py [M] [N] [K]
Prints, per stage of workgroup 8 / wave 0 of an unsplit launch: cycles waiting for the stage's weights (vmcnt), at the barrier (slab + the other waves),
in the stage's reads + MFMAs, and between the last MFMA and the next stage (fold, loop).  s_memtime ticks = shader cycles."""
import sys
import torch
from revisionllm_amd import hip, ops

lib = hip.lib()
dev = torch.device("cuda:0")
_a = [a for a in sys.argv[1:] if not a.startswith("--")]
M = int(_a[0]) if len(_a) > 0 else 140
N = int(_a[1]) if len(_a) > 1 else 12288
K = int(_a[2]) if len(_a) > 2 else 4096
mbp = ops.xp_blocks(M)
planes = torch.zeros(lib.rv_gemm_rows_ws_bytes(), dtype=torch.uint8, device=dev)
arrive = torch.zeros(4096, dtype=torch.int32, device=dev)
nw = max(2, int(8e8 // (N * K * 2)))
ws = [torch.randn(N * K // 2, device=dev).view(torch.int32) for _ in range(nw)]
x = (torch.randn(mbp * 16 * K, device=dev) * 0.1).to(hip.op_dtype())
c = torch.empty(M, N, device=dev)
runs = []
for i in range(12):
    rc = lib.rv_gemm_rows(hip.ptr(x), hip.ptr(ws[i % nw]), None, hip.ptr(c), M, N, K, hip.ptr(planes), hip.ptr(arrive), hip.RV_ACT_NONE, hip.RV_F32, hip.stream())
    assert rc == 0, hip.last_error()
    torch.cuda.synchronize()
    t = planes[:2048].view(torch.int64).view(64, 4).cpu()
    runs.append(t)
t = runs[-1]
n = int((t[:, 0] != 0).sum())
if "--brief" in sys.argv:
    import statistics
    w = [int(t[g, 1] - t[g, 0]) for g in range(1, n)]
    b = [int(t[g, 2] - t[g, 1]) for g in range(1, n)]
    m = [int(t[g, 3] - t[g, 2]) for g in range(1, n)]
    tl = [int(t[g + 1, 0] - t[g, 3]) for g in range(1, n - 1)]
    span = int(t[n - 1, 3]) - int(t[1, 0])
    print(f"stages {n}: per stage median wait {statistics.median(w)} barrier {statistics.median(b)} reads+mfma {statistics.median(m)} (mean {sum(m) / len(m):.0f}) "
          f"tail {statistics.median(tl)} (mean {sum(tl) / len(tl):.0f});  span {span / (n - 1):.0f} cycles per stage")
    sys.exit(0)
print(f"M={M} N={N} K={K}: {n} stamped stages (launch 12 of 12; cold weights every launch)")
print("stage   wait_weights   barrier   reads+mfma   tail_to_next")
tot = [0, 0, 0, 0]
for g in range(n):
    a, b, cc, d = [int(v) for v in t[g]]
    nxt = int(t[g + 1, 0]) - d if g + 1 < n else 0
    print(f"{g:5d} {b - a:12d} {cc - b:9d} {d - cc:12d} {nxt:12d}")
    for k, v in enumerate((b - a, cc - b, d - cc, nxt)):
        tot[k] += v
span = int(t[n - 1, 3]) - int(t[0, 0])
print("sum   ", *[f"{v:12d}" for v in tot], f"  span {span} cycles = {span / n:.0f} per stage")
med = torch.stack([r[:n] for r in runs[2:]]).float()
per = torch.stack([med[:, :, 1] - med[:, :, 0], med[:, :, 2] - med[:, :, 1], med[:, :, 3] - med[:, :, 2]], -1).median(0).values.sum(0)
print("median over 10 launches, summed over the stages: wait_weights %.0f  barrier %.0f  reads+mfma %.0f" % tuple(per.tolist()))
