"""One batched prefill pass of the headline (G prefills x (32 shared-prefix rows + 7 x 139 rows) = 4020 rows at G = 4) on an 8-block
Vicuna-7B-shaped model, a few times: run under rocprofv3 --kernel-trace --stats for per-kernel times at a FIXED row count.
  python tools/prefill_prof.py [G] [variant]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import engine
from revisionllm_amd.utils import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shape = synth.LlamaShape(layers=8, vocab=32000)
eng = engine.Engine(shape, adapter_text=False, device="cuda:0")
eng.init_synthetic(seed=3, llm=True, clip=False)
for a in sys.argv[2:]:
    k, v = a.split("=")
    eng.set_option(k, int(v))
B, P0, S, Smax, D = 7, 32, 139, 192, 4096
R = G * B
pool, _ = eng.new_kv_pool(R, Smax)
h = torch.randn(G * (P0 + B * S), D, device="cuda:0") * 0.02
for it in range(6):
    if it == 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    eng.llm_prefill_pool_groups(h.clone(), G, B, P0, pool, R, [B * g for g in range(G)], Smax)
torch.cuda.synchronize()
print(f"G={G}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per 8-block pass ({h.shape[0]} rows)")
