"""One COLD run of the GEMM shape that failed once in round 4 (test_gemm[bf16-1190-4096-512]): a fresh process, fresh device
context, output and stream-K workspace POISONED (NaN bit patterns) before the launch, M = 1190 (partial last row tile), the row-major-weight and
the fragment-packed forms of the ring kernels, both output types.  Prints one line: the sha256 of every result + its error against float64.
    python tools/gemm_cold.py [bf16|f16]          (tools/gemm_cold_loop.sh runs it N times and counts distinct lines)
Round 6: SHAPE=M,N,K in the environment runs that shape on the persistent stream-K ping-pong kernel instead (forced: gemm_tile_variant 5) - e.g. SHAPE=1005,22016,2048, whose
stream-K tail is cut into pieces of one, two and three k-tiles next to whole panels: the k-split main loop's prologue, peeled tiles and ring hazards from a cold start."""
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from revisionllm_amd import hip, ops  # noqa: E402
from helpers import feats  # noqa: E402

fl = sys.argv[1] if len(sys.argv) > 1 else "bf16"
hip.set_flavour(fl)
dt = hip.op_dtype()
M, N, K = 1190, 4096, 512
SK = os.environ.get("SHAPE")
if SK:
    M, N, K = (int(v) for v in SK.split(","))
dev = torch.device("cuda:0")
a = feats(f"gemm.a.{M}.{K}", (M, K), bf16=fl)
w = (feats(f"gemm.w.{N}.{K}", (N, K), bf16=fl) * (1.0 / math.sqrt(K))).to(dt).float()
ref = a.double() @ w.double().t()
ad, wd = a.to(dt).to(dev), w.to(dt).to(dev)
wpk = ops.pack_fragments(wd)
poison = float("nan")
out = []
for od in (dt, torch.float32):
    for packed in (False, True):
        y = torch.full((M, N), poison, dtype=od, device=dev)                 # every element must be overwritten
        ws = ops.stream_k_workspace(dev, fl)
        ws[16384:].fill_(0xFF)                                                # everything behind the hand-off header: NaN patterns
        if SK and not packed:
            continue
        ops.gemm(ad, wpk if packed else wd, out=y, w_packed=packed, stream_k=packed, ctx=hip.Options(flavour=fl, gemm_tile_variant=5) if SK else None)
        torch.cuda.synchronize()
        yc = y.float().cpu()
        err = float((yc.double() - ref).abs().max() / ref.abs().max())
        out.append("%s/%s %s err=%.3e" % ("packed" if packed else "rowmajor", "op16" if od == dt else "f32", hashlib.sha256(yc.numpy().tobytes()).hexdigest()[:12], err))
print(fl, " | ".join(out), flush=True)
