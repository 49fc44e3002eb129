"""The stage-2 adapter alone (ClipEncoder, text-conditioned, CLS out) on one recursion's input, N times: for `rocprofv3 --kernel-trace --stats`
(per-kernel time of the adapter when nothing else runs) and a HIP-event wall time.   python tools/adapter_prof.py [N] [f16|bf16] [option=value ...]
ADAPTER_GEOM=WxT[xQ] (environment): W windows of T frames, Q queries (default 100x256x1: one recursion; 32x1024x32 = the stage1_sparse windows in flight,
one query per window)."""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip  # noqa: E402
if len(sys.argv) > 2:
    hip.set_flavour(sys.argv[2])
from revisionllm_amd.model import ReVisionLlamaForCausalLM  # noqa: E402
from revisionllm_amd.utils import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
m = ReVisionLlamaForCausalLM(synth.LlamaShape(layers=1), device=dev)
m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, clip_adapter_text=True, clip_adapter_feature="cls", hierarchy=True,
                                                        adapter_input_dim=768, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None))
m.engine.init_synthetic(seed=0, llm=False, clip=True)
eng = m.engine
for a in sys.argv[3:]:          # option=value pairs, e.g. adapter_stream16=0
    k_, v_ = a.split("=")
    eng.set_option(k_, int(v_))
geom = [int(x) for x in os.environ.get("ADAPTER_GEOM", "100x256x1").split("x")]
Wn, Tn, Qn = geom[0], geom[1], (geom[2] if len(geom) > 2 else 1)
feats = torch.randn(Wn, Tn, 768, device=dev).to(hip.op_dtype())
qf = torch.randn(Qn, 16, 768, device=dev).to(hip.op_dtype())
mask = torch.ones(Qn, 16)
for _ in range(5):
    eng.clip_encoder(feats, qf, mask, "cls")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(n):
    eng.clip_encoder(feats, qf, mask, "cls")
b.record()
torch.cuda.synchronize()
print(f"adapter ({Wn} windows x {Tn} frames, {Qn} queries, {hip.flavour()} operands): {a.elapsed_time(b) / n:.3f} ms per recursion over {n} runs")
