"""Decode-step time (B = 7 rows, 171 cached positions, Vicuna-7B shapes)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from types import SimpleNamespace
from revisionllm_amd import hip, ops
from revisionllm_amd.model import ReVisionLlamaForCausalLM
from revisionllm_amd.utils import synth
dev = torch.device("cuda:0")
m = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device=dev)
m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, clip_adapter_text=True, clip_adapter_feature="cls",
                                                        hierarchy=True, adapter_input_dim=768, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None))
m.engine.init_synthetic(seed=0)
eng = m.engine
B, S = 7, 171
kv, Smax = eng.new_kv(B, S + 64)
h1 = torch.randn(B, 1, 4096, device=dev) * 0.02
def ev(fn, n=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rep in range(2):
    print(f"decode step {ev(lambda: eng.llm_forward(h1.clone(), S, kv, Smax)):.3f} ms", flush=True)
